#!/bin/bash
# On the GPU box: the GPU suite with 40-row tiles forced, then 32 rows against the per-pass choice on the bench configs.
set -u
OUT=gpurun_out/${1:-r03e}
mkdir -p $OUT
RND_TILE_HEIGHT=40 timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread > $OUT/gpu_tests_tall.txt 2>&1
rc=$?; tail -5 $OUT/gpu_tests_tall.txt
[ $rc -ne 0 ] && exit 1
for c in 1 2 4 0; do
  for th in 32 40 auto; do
    if [ $th = auto ]; then unset RND_TILE_HEIGHT; else export RND_TILE_HEIGHT=$th; fi
    timeout -k 10 150 python bench.py --config $c --no-cpu --repeats 2 > $OUT/line.json 2>> $OUT/bench.err || { echo "bench failed: $th config $c"; tail -5 $OUT/bench.err; exit 1; }
    python - $th $c $OUT/line.json <<'PY'
import json, sys
j = json.load(open(sys.argv[3]))
print("tile height %-5s cfg %s  %.5f ms  repeats %s  stages %s" % (sys.argv[1], sys.argv[2], j["ms_per_frame"], j["ms_per_frame_repeats"], j["stage_ms"]), flush=True)
PY
  done
done
