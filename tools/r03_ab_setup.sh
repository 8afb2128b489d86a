#!/bin/bash
# On the GPU box: the one-draw setup kernel (runs of 2, 4, 8 chunks per wave) against the general one, one frame in flight and three.
set -u
OUT=gpurun_out/${1:-r03l}
mkdir -p $OUT
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread > $OUT/gpu_tests.txt 2>&1 || { tail -5 $OUT/gpu_tests.txt; exit 1; }
tail -1 $OUT/gpu_tests.txt
for fif in 1 3; do for c in 1 4; do
  for v in plain default build/lib_run2.so build/lib_run8.so; do
    unset RND_SETUP_PLAIN RENDERER_AMD_LIB
    if [ $v = plain ]; then export RND_SETUP_PLAIN=1; elif [ $v != default ]; then export RENDERER_AMD_LIB=$PWD/$v; fi
    timeout -k 10 150 python bench.py --config $c --no-cpu --repeats 2 --frames-in-flight $fif > $OUT/line.json 2>> $OUT/bench.err || { echo "bench failed: $v config $c"; tail -5 $OUT/bench.err; exit 1; }
    python - "$v" $c $fif $OUT/line.json <<'PY'
import json, sys
j = json.load(open(sys.argv[4]))
print("%-22s cfg %s fif %s  %.5f ms  repeats %s  stages %s" % (sys.argv[1], sys.argv[2], sys.argv[3], j["ms_per_frame"], j["ms_per_frame_repeats"], j["stage_ms"]), flush=True)
PY
  done
done; done
