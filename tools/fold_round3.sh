#!/bin/bash
# On the CPU side, after gpurun merged the outputs of tools/collect_round3.sh TAG: folds them into profiles/TAG_*.
set -e
TAG=${1:-r03z}
python tools/summarize_profiles.py gpurun_out/prof_$TAG profiles/$TAG 1
python tools/summarize_profiles.py gpurun_out/prof_${TAG}_c2 profiles/${TAG}_c2 2
python tools/summarize_profiles.py gpurun_out/prof_${TAG}_c0 profiles/${TAG}_c0 0
python tools/summarize_profiles.py gpurun_out/prof_${TAG}_c4 profiles/${TAG}_c4 4
cp gpurun_out/prof_${TAG}_fif3/overlap.json profiles/${TAG}_overlap.json
cp gpurun_out/prof_${TAG}_c2_fif3/overlap.json profiles/${TAG}_c2_overlap.json
cp "$(ls gpurun_out/prof_${TAG}_fif3/trace/*/*kernel_stats.csv | head -1)" profiles/${TAG}_three_in_flight_kernel_stats.csv
for f in bench.json bench_driver_style.json bench_other_configs.jsonl bench_4_samples.jsonl band_rehearsal.jsonl tile_rehearsal.jsonl gpus2_gloo_one_gpu.json bands_one_rank_rccl.json gpu_tests.txt sequence_sweep.txt; do
  cp gpurun_out/$TAG/$f profiles/${TAG}_$f
done
