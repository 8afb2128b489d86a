#!/bin/bash
# On the GPU box: the benchmark configs with 2, 3 and 4 frames in flight.   tools/fif_sweep.sh OUTDIR "CONFIGS"
set -u
OUT=$1; CONFIGS=$2
mkdir -p $OUT
for c in $CONFIGS; do
  for f in 2 3 4 2 3; do
    timeout -k 10 150 python bench.py --config $c --no-cpu --repeats 2 --frames-in-flight $f > $OUT/line.json 2>> $OUT/bench.err || { echo "bench failed: config $c"; tail -5 $OUT/bench.err; exit 1; }
    python - $f $c $OUT/line.json <<'PY'
import json, sys
j = json.load(open(sys.argv[3]))
print("in flight %s cfg %s  %.5f ms  repeats %s" % (sys.argv[1], sys.argv[2], j["ms_per_frame"], j["ms_per_frame_repeats"]), flush=True)
PY
    echo "{\"frames_in_flight\": $f, \"config\": $c, \"line\": $(cat $OUT/line.json)}" >> $OUT/fif.jsonl
  done
done
