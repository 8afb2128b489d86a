set -e
mkdir -p gpurun_out/r02e
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread > gpurun_out/r02e/tests.log 2>&1; tail -2 gpurun_out/r02e/tests.log
timeout -k 10 200 python bench.py > gpurun_out/r02e/bench.json 2> gpurun_out/r02e/bench.err; cat gpurun_out/r02e/bench.json
for c in 0 2 3 4; do timeout -k 10 200 python bench.py --config $c --no-cpu >> gpurun_out/r02e/bench_other_configs.jsonl 2>> gpurun_out/r02e/bench.err; done
for c in 0 1 2; do timeout -k 10 200 python bench.py --config $c --samples 4 --no-cpu >> gpurun_out/r02e/bench_4_samples.jsonl 2>> gpurun_out/r02e/bench.err; done
bash tools/collect_profiles.sh r02e 1
bash tools/collect_profiles.sh r02e_c2 2
bash tools/collect_profiles.sh r02e_c0 0
