#!/bin/bash
# Runs on the GPU box (via gpurun): counter passes that say where a kernel's waves spend their cycles --
# lanes active per vector instruction, cycles per instruction class, the texture-address and LDS queues,
# the vector cache's stalls and latencies.  One rocprofv3 run per pass (no trace domains with --pmc).
#   bash tools/collect_issue.sh TAG CONFIG  ->  gpurun_out/issue_TAG/pmc_*
set -u
export TMPDIR=/tmp
OUT=gpurun_out/issue_${1:-run}
mkdir -p $OUT
B="python3 bench.py --steps 20 --warmup 5 --frames-per-step 1 --profile-steps 5 --no-cpu --frames-in-flight 1 --config ${2:-1}"
pass() { local name=$1; shift; timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc_$name -- $B > $OUT/pmc_$name.log 2>&1 || exit 1; }
pass a SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_ANY
pass b SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_BUSY_CYCLES TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE
pass c TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TD_TD_BUSY_sum TD_TC_STALL_sum
pass d TA_TOTAL_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TOTAL_READ_sum SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS
pass e SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_BRANCH
echo collected $OUT
