set -e
mkdir -p gpurun_out/r02f
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread > gpurun_out/r02f/tests.log 2>&1; tail -2 gpurun_out/r02f/tests.log
timeout -k 10 200 python bench.py > gpurun_out/r02f/bench.json 2> gpurun_out/r02f/bench.err; cat gpurun_out/r02f/bench.json
for c in 0 2 3 4; do timeout -k 10 200 python bench.py --config $c --no-cpu >> gpurun_out/r02f/bench_other_configs.jsonl 2>> gpurun_out/r02f/bench.err; done
for c in 0 1 2; do timeout -k 10 200 python bench.py --config $c --samples 4 --no-cpu >> gpurun_out/r02f/bench_4_samples.jsonl 2>> gpurun_out/r02f/bench.err; done
bash tools/collect_profiles.sh r02f 1
bash tools/collect_profiles.sh r02f_c2 2
bash tools/collect_profiles.sh r02f_c0 0
