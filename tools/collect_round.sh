set -e
TAG=${1:-r02h}
mkdir -p gpurun_out/$TAG
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread > gpurun_out/$TAG/tests.log 2>&1; tail -2 gpurun_out/$TAG/tests.log
timeout -k 10 200 python bench.py > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err; cat gpurun_out/$TAG/bench.json
timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu > gpurun_out/$TAG/bench_driver_style.json 2>> gpurun_out/$TAG/bench.err; cat gpurun_out/$TAG/bench_driver_style.json
for c in 0 2 3 4; do timeout -k 10 200 python bench.py --config $c --no-cpu >> gpurun_out/$TAG/bench_other_configs.jsonl 2>> gpurun_out/$TAG/bench.err; done
for c in 0 1 2; do timeout -k 10 200 python bench.py --config $c --samples 4 --no-cpu >> gpurun_out/$TAG/bench_4_samples.jsonl 2>> gpurun_out/$TAG/bench.err; done
bash tools/collect_profiles.sh $TAG 1
bash tools/collect_profiles.sh ${TAG}_c2 2
bash tools/collect_profiles.sh ${TAG}_c0 0
