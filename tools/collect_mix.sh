set -u
export TMPDIR=/tmp
export RENDERER_AMD_LIB=build/lib_probes.so
for m in 0 1 2; do
  export RND_DEBUG=$m
  OUT=gpurun_out/mix_c${1}_m$m; mkdir -p $OUT
  B="python3 bench.py --steps 20 --warmup 5 --frames-per-step 1 --profile-steps 5 --no-cpu --frames-in-flight 1 --config $1"
  timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT --output-format csv -d $OUT/pmc_a -- $B > $OUT/a.log 2>&1 || exit 1
  timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_TRANS_F32 SQ_WAVES --output-format csv -d $OUT/pmc_b -- $B > $OUT/b.log 2>&1 || exit 1
done
echo ok
