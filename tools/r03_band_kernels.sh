#!/bin/bash
# On the GPU box: per-kernel durations (rocprofv3 kernel trace) of the middle band of 8 of the 8K frame, one frame in flight,
# for each value of an environment variable.
#   tools/r03_band_kernels.sh OUTDIR VAR "VALUES"      (value "-" = the variable unset)
set -u
OUT=gpurun_out/$1; VAR=$2; VALUES=$3
mkdir -p $OUT
i=0
for v in $VALUES; do
  i=$((i + 1))
  if [ "$v" = "-" ]; then unset $VAR; else export $VAR=$v; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace$i -- python3 bench.py --config 3 --mode bands --one-shard-of 8 --shard-rank 4 --frames-in-flight 1 --no-cpu --steps 100 --warmup 20 --frames-per-step 1 > $OUT/log$i.txt 2>&1 || { echo "failed: $v"; tail -3 $OUT/log$i.txt; exit 1; }
  echo "== $VAR=$v"
  head -5 $(ls $OUT/trace$i/*/*kernel_stats.csv | head -1) | tail -4 | cut -d, -f1,4 | cut -c1-110
done
