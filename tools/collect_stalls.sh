#!/bin/bash
# Runs on the GPU box (via gpurun): counter passes that say where waves wait (instruction fetch, vector-memory
# and LDS queues, texture-address unit, vector L1), one rocprofv3 run per counter group, no trace domains.
#   tools/collect_stalls.sh NAME [bench args...]   -> gpurun_out/stalls_NAME/
set -u
export TMPDIR=/tmp
NAME=${1:-run}; shift
OUT=gpurun_out/stalls_$NAME
mkdir -p $OUT
B="python3 bench.py --steps 20 --warmup 5 --frames-per-step 1 --profile-steps 5 --no-cpu --frames-in-flight 1 $*"
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_a -- $B > $OUT/pmc_a.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_MISSES SQC_TC_STALL SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc_b -- $B > $OUT/pmc_b.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_c -- $B > $OUT/pmc_c.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/pmc_d -- $B > $OUT/pmc_d.log 2>&1 || exit 1
python3 tools/summarize_profiles.py $OUT $OUT/sum > /dev/null
echo collected $OUT
