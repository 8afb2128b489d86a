#!/bin/bash
# On the GPU box: how many workgroups should walk the lists of visible chunks (RND_LIST_WALK_BLOCKS), against RND_COMPACT=0:
# whole frames of configs 2 and 3, and the middle band of 8 of the 8K frame.
#   tools/r03_walk.sh OUTDIR "VALUES"
set -u
OUT=gpurun_out/$1; VALUES=$2
mkdir -p $OUT
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread > $OUT/gpu_tests.txt 2>&1
rc=$?; tail -3 $OUT/gpu_tests.txt
[ $rc -ne 0 ] && exit 1
run() { # label, then bench args
  local label=$1; shift
  timeout -k 10 200 python bench.py --no-cpu "$@" > $OUT/line.json 2>> $OUT/bench.err || { echo "bench failed: $label"; tail -5 $OUT/bench.err; exit 1; }
  python - "$label" $OUT/line.json <<'PY'
import json, sys
j = json.load(open(sys.argv[2]))
print("%-34s %.5f ms  %s  stages %s" % (sys.argv[1], j["ms_per_frame"], j.get("ms_per_frame_repeats"), j.get("stage_ms")), flush=True)
PY
}
for v in off $VALUES; do
  if [ "$v" = "off" ]; then export RND_COMPACT=0; unset RND_LIST_WALK_BLOCKS; else unset RND_COMPACT; export RND_LIST_WALK_BLOCKS=$v; fi
  run "walk=$v cfg 2" --config 2 --repeats 2
  run "walk=$v cfg 3" --config 3 --repeats 2
  run "walk=$v band 4 of 8" --config 3 --mode bands --one-shard-of 8 --shard-rank 4 --frames-in-flight 1 --steps 100 --warmup 20 --frames-per-step 1
done
