#!/bin/bash
# Runs on the GPU box (via gpurun): kernel trace + the two SQ counter passes only, then prints
# the per-kernel table.  Usage: tools/collect_sq.sh NAME [bench args...]
set -u
export TMPDIR=/tmp
NAME=${1:-run}; shift
OUT=gpurun_out/sq_$NAME
mkdir -p $OUT
B="python3 bench.py --steps 20 --warmup 5 --frames-per-step 1 --profile-steps 5 --no-cpu --frames-in-flight 1 $*"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B > $OUT/trace.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_sq -- $B > $OUT/pmc_sq.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/pmc_sq2 -- $B > $OUT/pmc_sq2.log 2>&1 || exit 1
python3 tools/summarize_profiles.py $OUT $OUT/sum > /dev/null
python3 - $OUT/sum_pmc.csv <<'PY'
import csv, sys
want = ["avg_us", "SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_WAIT_INST_ANY", "SQ_INST_CYCLES_VMEM", "vgpr", "scratch"]
for r in csv.DictReader(open(sys.argv[1])):
    if not r.get("avg_us"): continue
    print(r["kernel"][:28], " ".join("%s=%s" % (k.replace("SQ_", ""), ("%.0f" % float(r[k])) if r.get(k) and k != "avg_us" else r.get(k, "")[:6]) for k in want))
PY
