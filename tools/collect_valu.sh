#!/bin/bash
# Runs on the GPU box: instruction counts per kernel for one configuration (one counter pass).
#   bash tools/collect_valu.sh TAG CONFIG   (the library comes from RENDERER_AMD_LIB as usual)
set -u
export TMPDIR=/tmp
OUT=gpurun_out/valu_${1:-run}
mkdir -p $OUT
B="python3 bench.py --steps 20 --warmup 5 --frames-per-step 1 --profile-steps 5 --no-cpu --frames-in-flight 1 --config ${2:-1}"
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_a -- $B > $OUT/a.log 2>&1 || exit 1
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B > $OUT/trace.log 2>&1 || exit 1
echo collected $OUT
