#!/bin/bash
# On the GPU box: parity tests of the in-tree build, then an A/B of library builds on the benchmark configs.
#   tools/ab_round.sh OUTDIR "CONFIGS" LIB [LIB...]    ("default" = the in-tree build; SKIP_TESTS=1: the A/B alone)
set -u
OUT=$1; CONFIGS=$2; shift 2
mkdir -p $OUT
if [ -z "${SKIP_TESTS:-}" ]; then
  timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread > $OUT/tests.log 2>&1
  rc=$?; tail -4 $OUT/tests.log
  [ $rc -ne 0 ] && exit 1
fi
for c in $CONFIGS; do
  for lib in "$@"; do
    for rep in 1 2; do
      if [ "$lib" = default ]; then unset RENDERER_AMD_LIB; else export RENDERER_AMD_LIB=$PWD/$lib; fi
      timeout -k 10 150 python bench.py --config $c --no-cpu --repeats 2 > $OUT/line.json 2>> $OUT/bench.err || { echo "bench failed: $lib config $c"; tail -5 $OUT/bench.err; exit 1; }
      python - "$lib" $c $OUT/line.json <<'PY'
import json, sys
j = json.load(open(sys.argv[3]))
print("%-28s cfg %s  %.5f ms  repeats %s  stages %s" % (sys.argv[1], sys.argv[2], j["ms_per_frame"], j["ms_per_frame_repeats"], j["stage_ms"]), flush=True)
PY
      echo "{\"lib\": \"$lib\", \"config\": $c, \"line\": $(cat $OUT/line.json)}" >> $OUT/ab.jsonl
    done
  done
done
