#!/bin/bash
# On the GPU box: frames in flight against the tile kernel's workgroups per CU (RND_TILE_WORKGROUPS, unset = measured).
set -u
OUT=gpurun_out/fif3; mkdir -p $OUT
run() { # config, frames in flight, workgroups
  if [ "$3" = measured ]; then unset RND_TILE_WORKGROUPS; else export RND_TILE_WORKGROUPS=$3; fi
  timeout -k 10 150 python bench.py --config $1 --no-cpu --repeats 2 --frames-in-flight $2 > $OUT/line.json 2>> $OUT/bench.err || { echo "bench failed"; tail -5 $OUT/bench.err; exit 1; }
  python - "cfg $1 in flight $2 workgroups $3" $OUT/line.json <<'PY'
import json, sys
j = json.load(open(sys.argv[2]))
print("%-44s %.5f ms  repeats %s" % (sys.argv[1], j["ms_per_step"], j["ms_per_step_repeats"]), flush=True)
PY
}
for c in 1 0 4 2; do for f in 3 4; do for w in measured 4 5; do run $c $f $w; done; done; done
