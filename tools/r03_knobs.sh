#!/bin/bash
# On the GPU box: the GPU suite once under each of the library's A/B settings (every path they select renders the same bits).
#   tools/r03_knobs.sh OUTDIR
set -u
OUT=gpurun_out/$1
mkdir -p $OUT
i=0
for setting in "RND_COMPACT=0" "RND_TILE_GENERIC=1" "RND_TILE_HEIGHT=40" "RND_TILE_HEIGHT=32" "RND_LIST_WALK_BLOCKS=3" "RND_CMD_STREAMS=1" "RND_STREAM_PRIORITY=normal" "RND_TILE_WORKGROUPS=4"; do
  i=$((i + 1))
  env $setting timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread > $OUT/tests_$i.txt 2>&1
  rc=$?
  echo "$setting: exit $rc: $(tail -1 $OUT/tests_$i.txt)"
  [ $rc -ne 0 ] && { tail -20 $OUT/tests_$i.txt; exit 1; }
done
echo "all settings green"
