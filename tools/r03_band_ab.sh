#!/bin/bash
# On the GPU box: one environment setting A/B on the whole frames of configs 2 and 3 and on the middle band of 8 of the 8K frame.
#   tools/r03_band_ab.sh OUTDIR VAR "VALUES"      (value "-" = the variable unset)
set -u
OUT=gpurun_out/$1; VAR=$2; VALUES=$3
mkdir -p $OUT
run() {
  local label=$1; shift
  timeout -k 10 200 python bench.py --no-cpu "$@" > $OUT/line.json 2>> $OUT/bench.err || { echo "bench failed: $label"; tail -5 $OUT/bench.err; exit 1; }
  python - "$label" $OUT/line.json <<'PY'
import json, sys
j = json.load(open(sys.argv[2]))
print("%-44s %.5f ms  %s  stages %s" % (sys.argv[1], j["ms_per_frame"], j.get("ms_per_frame_repeats"), j.get("stage_ms")), flush=True)
PY
}
for v in $VALUES; do
  if [ "$v" = "-" ]; then unset $VAR; else export $VAR=$v; fi
  run "$VAR=$v cfg 2" --config 2 --repeats 2
  run "$VAR=$v cfg 3" --config 3 --repeats 2
  run "$VAR=$v band 4 of 8" --config 3 --mode bands --one-shard-of 8 --shard-rank 4 --frames-in-flight 1 --steps 100 --warmup 20 --frames-per-step 1
done
