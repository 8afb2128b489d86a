#!/bin/bash
# On the GPU box: GPU suite, then the four-sample frames of configs 0, 1, 2.
#   tools/r03_msaa.sh OUTDIR
set -u
OUT=gpurun_out/$1
mkdir -p $OUT
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread > $OUT/gpu_tests.txt 2>&1
rc=$?; tail -5 $OUT/gpu_tests.txt
[ $rc -ne 0 ] && exit 1
for c in 0 1 2; do
  timeout -k 10 200 python bench.py --config $c --samples 4 --no-cpu --repeats 2 > $OUT/line.json 2>> $OUT/bench.err || { echo "bench failed"; tail -5 $OUT/bench.err; exit 1; }
  python - $c $OUT/line.json <<'PY'
import json, sys
j = json.load(open(sys.argv[2]))
print("4 samples cfg %s  %.5f ms  repeats %s  stages %s" % (sys.argv[1], j["ms_per_frame"], j["ms_per_frame_repeats"], j["stage_ms"]), flush=True)
PY
done
