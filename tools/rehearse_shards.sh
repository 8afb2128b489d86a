#!/bin/bash
# Runs on the GPU box: one GPU renders the share of rank n/2 of n = 2, 4, 8 of config 3 (8K), as bands and as
# interleaved tiles (one frame in flight, no exchange), then the two-rank launch of bench.py over gloo on the one GPU.
#   bash tools/rehearse_shards.sh TAG  ->  gpurun_out/TAG/{band,tile}_rehearsal.jsonl, gpus2_gloo_one_gpu.json
set -u
OUT=gpurun_out/${1:-rehearsal}
mkdir -p $OUT
for n in 2 4 8; do
  timeout -k 10 200 python bench.py --config 3 --mode bands --one-shard-of $n --shard-rank $((n / 2)) --frames-in-flight 1 --no-cpu --steps 100 --warmup 20 --frames-per-step 1 >> $OUT/band_rehearsal.jsonl 2>> $OUT/err.log || exit 1
  timeout -k 10 200 python bench.py --config 3 --one-shard-of $n --shard-rank $((n / 2)) --frames-in-flight 1 --no-cpu --steps 100 --warmup 20 --frames-per-step 1 >> $OUT/tile_rehearsal.jsonl 2>> $OUT/err.log || exit 1
done
BENCH_BACKEND=gloo timeout -k 10 400 python bench.py --gpus 2 --steps 200 --warmup 50 --frames-per-step 1 > $OUT/gpus2_gloo_one_gpu.json 2>> $OUT/err.log || exit 1
echo done $OUT
