#!/bin/bash
# On the GPU box: frames in flight 2..5 with as many command-buffer streams, configs 1 and 2.
set -u
OUT=gpurun_out/${1:-r03n}; mkdir -p $OUT
for c in 1 2; do for fif in 2 3 4 5; do
  RND_CMD_STREAMS=$fif timeout -k 10 150 python bench.py --config $c --no-cpu --repeats 2 --frames-in-flight $fif > $OUT/line.json 2>> $OUT/bench.err || { echo failed; exit 1; }
  python - $fif $c $OUT/line.json <<'PY'
import json, sys
j = json.load(open(sys.argv[3]))
print("frames in flight %s cfg %s  %.5f ms  repeats %s" % (sys.argv[1], sys.argv[2], j["ms_per_frame"], j["ms_per_frame_repeats"]), flush=True)
PY
done; done
