#!/bin/bash
# On the GPU box: the randomised parity sweeps against the oracle, with seeds no earlier sweep used.
#   bash tools/r03_fuzz.sh OUTDIR FIRST_SEED [SCALE]      (SCALE multiplies the case counts; default 1)
set -u
OUT=gpurun_out/$1; FIRST=$2; K=${3:-1}
mkdir -p $OUT
run() { # name, time limit, then the command
  local name=$1 limit=$2; shift 2
  timeout -k 10 $limit "$@" > $OUT/$name.txt 2>&1
  local rc=$?
  echo "$name: exit $rc: $(tail -1 $OUT/$name.txt)"
  [ $rc -ne 0 ] && exit 1
  return 0
}
run soup 900 python -u tests/fuzz_soup.py $((2000 * K)) $FIRST
run cull 900 python -u tests/fuzz_cull.py $((1000 * K)) $FIRST
run state 900 python -u tests/fuzz_state.py $((600 * K)) $FIRST
run clip 600 python -u tests/fuzz_clip.py $((400 * K)) $FIRST
run shards 600 python -u tests/fuzz_shards.py $((300 * K)) $FIRST
run noalloc 600 python -u tests/fuzz_noalloc.py $((300 * K)) $FIRST
run large 900 python -u tests/fuzz_large.py $(seq $FIRST $((FIRST + 30 * K - 1)))
RND_DEBUG_SYNC=1 run sequence 900 python -u tests/fuzz_sequence.py $((60 * K)) $FIRST
echo "all sweeps clean"
