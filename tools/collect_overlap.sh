#!/bin/bash
# On the GPU box: a rocprofv3 kernel trace of the bench command as the driver runs it (three frames in flight), for
# tools/overlap_from_trace.py.  The program itself follows `--` (no wrapper).
set -u
export TMPDIR=/tmp
OUT=gpurun_out/prof_${1:-run}_fif3
mkdir -p $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --gpus 1 --steps 8 --warmup 2 --no-cpu --profile-steps 0 --repeats 0 --config ${2:-1} > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
python3 tools/overlap_from_trace.py $OUT/trace $OUT/overlap.json 300
