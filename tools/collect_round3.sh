#!/bin/bash
# On the GPU box: everything the round's tables come from, for the in-tree build.
#   bash tools/collect_round3.sh TAG   ->  gpurun_out/TAG/*, gpurun_out/prof_TAG*/ (folded on the CPU side by tools/fold_round3.sh TAG)
set -u
TAG=${1:-r03z}
OUT=gpurun_out/$TAG
mkdir -p $OUT
step() { echo "== $*" | tee -a $OUT/progress.txt; }
step tests
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread > $OUT/gpu_tests.txt 2>&1 || { tail -5 $OUT/gpu_tests.txt; exit 1; }
tail -2 $OUT/gpu_tests.txt
step bench
timeout -k 10 200 python bench.py > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_style.json 2>> $OUT/bench.err || exit 1
for c in 0 2 3 4; do timeout -k 10 200 python bench.py --config $c --no-cpu >> $OUT/bench_other_configs.jsonl 2>> $OUT/bench.err || exit 1; done
for c in 0 1 2; do timeout -k 10 200 python bench.py --config $c --samples 4 --no-cpu >> $OUT/bench_4_samples.jsonl 2>> $OUT/bench.err || exit 1; done
step profiles
bash tools/collect_profiles.sh $TAG 1 || exit 1
bash tools/collect_profiles.sh ${TAG}_c2 2 || exit 1
bash tools/collect_profiles.sh ${TAG}_c0 0 || exit 1
bash tools/collect_profiles.sh ${TAG}_c4 4 || exit 1
step overlap
bash tools/collect_overlap.sh $TAG 1 || exit 1
bash tools/collect_overlap.sh ${TAG}_c2 2 || exit 1
step rehearsal
bash tools/rehearse_shards.sh $TAG || exit 1
bash tools/rccl_one_rank.sh $OUT || exit 1
step sequences
RND_DEBUG_SYNC=1 timeout -k 10 400 python -u tests/fuzz_sequence.py 20 20 > $OUT/sequence_sweep.txt 2>&1 || { tail -3 $OUT/sequence_sweep.txt; exit 1; }
tail -1 $OUT/sequence_sweep.txt
step done
