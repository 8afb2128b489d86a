# Runs on the GPU box: the same long soup sweep on several builds of the library.  bash tools/fuzz_bisect.sh CASES FIRST LIB...
CASES=$1; FIRST=$2; shift 2
mkdir -p gpurun_out/bisect
for lib in "$@"; do
  if [ "$lib" = default ]; then unset RENDERER_AMD_LIB; else export RENDERER_AMD_LIB=$PWD/$lib; fi
  n=$(basename $lib .so)
  timeout -k 10 400 python -u tests/fuzz_soup.py $CASES $FIRST > gpurun_out/bisect/$n.log 2>&1
  echo "$lib: exit $? : $(tail -1 gpurun_out/bisect/$n.log | cut -c1-160) : last seed $(grep -c . gpurun_out/bisect/$n.log)"
done
