# Runs on the GPU box (via gpurun): the three randomised parity sweeps with seeds no earlier sweep used.
#   bash tools/fuzz_round.sh TAG SOUP_CASES SOUP_FIRST CULL_CASES CULL_FIRST LARGE_FIRST LARGE_COUNT
set -e
TAG=${1:-fuzz}
mkdir -p gpurun_out/$TAG
timeout -k 10 900 python -u tests/fuzz_soup.py ${2:-2000} ${3:-10000} 2>&1 | tee gpurun_out/$TAG/soup.log | tail -3
timeout -k 10 600 python -u tests/fuzz_cull.py ${4:-500} ${5:-5000} 2>&1 | tee gpurun_out/$TAG/cull.log | tail -3
timeout -k 10 600 python -u tests/fuzz_large.py $(seq ${6:-100} $((${6:-100} + ${7:-30} - 1))) 2>&1 | tee gpurun_out/$TAG/large.log | tail -3
