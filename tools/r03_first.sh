#!/bin/bash
# On the GPU box: GPU suite of the in-tree build, the sequence sweep under the default stream pool and under a pool of 16, the driver-style bench line.
set -u
OUT=gpurun_out/${1:-r03a}
mkdir -p $OUT
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread > $OUT/gpu_tests.txt 2>&1
rc=$?; tail -5 $OUT/gpu_tests.txt
[ $rc -ne 0 ] && exit 1
RND_DEBUG_SYNC=1 timeout -k 10 400 python -u tests/fuzz_sequence.py 14 0 > $OUT/seq_default.log 2>&1; echo "sequence sweep, default pool: exit $?"; tail -3 $OUT/seq_default.log
RND_CMD_STREAMS=16 RND_DEBUG_SYNC=1 timeout -k 10 400 python -u tests/fuzz_sequence.py 14 0 > $OUT/seq_pool16.log 2>&1; echo "sequence sweep, pool of 16: exit $?"; tail -3 $OUT/seq_pool16.log
timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_style.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
cat $OUT/bench_driver_style.json
