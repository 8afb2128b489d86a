# Runs on the GPU box: the engine clock the device reports while the bench's frames run (sampled once a second).
#   bash tools/probe_clock.sh CONFIG
mkdir -p gpurun_out/micro
OUT=gpurun_out/micro/clock_config${1:-1}.txt
: > $OUT
timeout -k 10 150 python bench.py --config ${1:-1} --steps 4000 --warmup 5 --repeats 0 --no-cpu --profile-steps 1 > gpurun_out/micro/clock_bench_config${1:-1}.json 2> gpurun_out/micro/clock_bench.err &
B=$!
for i in $(seq 1 14); do
  sleep 1
  echo "t=$i" >> $OUT
  timeout 10 rocm-smi --showclocks --showpower --showuse 2>&1 | grep -E "sclk|fclk|mclk|Power|GPU use" >> $OUT
done
wait $B
echo "bench exit $?" >> $OUT
cat $OUT | tail -60
