#!/bin/bash
# On the GPU box: GPU suite, then the dense lists of visible chunks on (default) and off (RND_COMPACT=0):
# whole frames of configs 2 and 3, and one rank's band of 2, 4, 8 of the 8K frame.
#   tools/r03_compact.sh OUTDIR
set -u
OUT=gpurun_out/$1
mkdir -p $OUT
if [ -z "${SKIP_TESTS:-}" ]; then
  timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread > $OUT/gpu_tests.txt 2>&1
  rc=$?; tail -5 $OUT/gpu_tests.txt
  [ $rc -ne 0 ] && exit 1
fi
SKIP_TESTS=1 bash tools/r03_ab_env.sh $1 "2 3" RND_COMPACT "0 -" || exit 1
for v in 0 -; do
  if [ "$v" = "-" ]; then unset RND_COMPACT; else export RND_COMPACT=$v; fi
  for n in 2 4 8; do
    for rank in 0 $((n / 2)); do
      timeout -k 10 200 python bench.py --config 3 --mode bands --one-shard-of $n --shard-rank $rank --frames-in-flight 1 --no-cpu --steps 100 --warmup 20 --frames-per-step 1 > $OUT/line.json 2>> $OUT/bench.err || { echo "rehearsal failed"; tail -5 $OUT/bench.err; exit 1; }
      python - "RND_COMPACT=$v" $n $rank $OUT/line.json <<'PY'
import json, sys
j = json.load(open(sys.argv[4]))
print("%-16s band %s of %s  %.5f ms  stages %s" % (sys.argv[1], sys.argv[3], sys.argv[2], j["ms_per_frame"], j.get("stage_ms")), flush=True)
PY
    done
  done
done
