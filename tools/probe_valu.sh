#!/bin/bash
# VALU instruction count and duration of the frame's kernels under RND_DEBUG probe masks (GPU box).
# Needs a library built with -DRND_PROBES (EXTRA=-DRND_PROBES tools/build_variant.sh probes) named by RENDERER_AMD_LIB.
export TMPDIR=/tmp
for m in "$@"; do
  OUT=gpurun_out/valu_$m; mkdir -p $OUT
  RND_DEBUG=$m timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $OUT -- python3 bench.py --steps 10 --warmup 3 --frames-per-step 1 --profile-steps 3 --no-cpu --frames-in-flight 1 > $OUT/log.txt 2>&1 || exit 1
  python3 - $OUT $m <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    if "k_shade" in k or "k_tile" in k or "k_setup" in k or "k_vertex" in k:
        print("mask", sys.argv[2], k.split("(")[0][-14:], " ".join("%s=%.0f" % (n.replace("SQ_INSTS_", ""), sum(v) / len(v)) for n, v in sorted(c.items())))
PY
done
