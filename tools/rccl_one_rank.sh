# Runs on the GPU box: the band exchange over RCCL with one rank (the exchange code of bench.py --mode bands on real RCCL),
# and the process's exit status.   bash tools/rccl_one_rank.sh OUTDIR
OUT=${1:-gpurun_out/rccl_one_rank}
mkdir -p $OUT
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --mode bands --config 3 --steps 20 --warmup 5 --no-cpu > $OUT/stdout.txt 2> $OUT/err.log
rc=$?
grep '^{"metric"' $OUT/stdout.txt > $OUT/bands_one_rank_rccl.json
echo "exit status $rc" | tee $OUT/exit_status.txt
exit $rc
