#!/bin/bash
# On the GPU box: the GPU suite of the in-tree build, then an A/B of one environment setting on the bench configs.
#   tools/r03_ab_env.sh OUTDIR "CONFIGS" VAR "VALUES" [extra bench args]      (value "-" = the variable unset)
set -u
OUT=gpurun_out/$1; CONFIGS=$2; VAR=$3; VALUES=$4; shift 4
mkdir -p $OUT
if [ -z "${SKIP_TESTS:-}" ]; then
  timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread > $OUT/gpu_tests.txt 2>&1
  rc=$?; tail -5 $OUT/gpu_tests.txt
  [ $rc -ne 0 ] && exit 1
fi
for c in $CONFIGS; do
  for v in $VALUES; do
    for rep in 1 2; do
      if [ "$v" = "-" ]; then unset $VAR; else export $VAR=$v; fi
      timeout -k 10 150 python bench.py --config $c --no-cpu --repeats 2 "$@" > $OUT/line.json 2>> $OUT/bench.err || { echo "bench failed: $VAR=$v config $c"; tail -5 $OUT/bench.err; exit 1; }
      python - "$VAR=$v" $c $OUT/line.json <<'PY'
import json, sys
j = json.load(open(sys.argv[3]))
print("%-22s cfg %s  %.5f ms  repeats %s  stages %s" % (sys.argv[1], sys.argv[2], j["ms_per_frame"], j["ms_per_frame_repeats"], j["stage_ms"]), flush=True)
PY
    done
  done
done
