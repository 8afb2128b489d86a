#!/bin/bash
# On the GPU box: the read-counter calibration (tools/micro/fetch_calib.hip, built beforehand as build/fetch_calib)
# under rocprofv3, one run per counter group, folded into gpurun_out/fetch_calib/fetch_calib.json.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/fetch_calib
mkdir -p $OUT
timeout -k 10 120 build/fetch_calib > $OUT/plain.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- build/fetch_calib > $OUT/pmc_fetch.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $OUT/pmc_req -- build/fetch_calib > $OUT/pmc_req.log 2>&1 || exit 1
python3 - $OUT <<'PY'
import collections, csv, glob, json, re, sys
out = sys.argv[1]
known, ms = {}, {}
for line in open(out + "/plain.log"):
    m = re.match(r"(\S+)\s+known_bytes (\d+)\s+ms ([\d.]+)", line)
    if m:
        known[m.group(1)] = float(m.group(2)); ms[m.group(1)] = float(m.group(3))  # the second repetition overwrites the first
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ("pmc_fetch", "pmc_req"):
    for f in glob.glob(out + "/" + sub + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            name = {"sparse<4>": "sparse4", "sparse<16>": "sparse16", "sparse<12>": "sparse12"}.get(name, name)
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for name in known:
    c = {k: sum(v) / len(v) for k, v in acc.get(name, {}).items()}
    e = {"known_bytes": known[name], "ms": ms[name], "counters": c}
    if "FETCH_SIZE" in c:
        e["fetch_size_bytes"] = c["FETCH_SIZE"] * 1024.0
        e["known_over_fetch_size"] = known[name] / (c["FETCH_SIZE"] * 1024.0)
    if "TCC_EA0_RDREQ_128B_sum" in c:
        e["bytes_by_request_size"] = 32.0 * c.get("TCC_EA0_RDREQ_32B_sum", 0) + 64.0 * c.get("TCC_EA0_RDREQ_64B_sum", 0) + 128.0 * c["TCC_EA0_RDREQ_128B_sum"]
        e["known_over_bytes_by_request_size"] = known[name] / e["bytes_by_request_size"] if e["bytes_by_request_size"] else None
    res[name] = e
json.dump(res, open(out + "/fetch_calib.json", "w"), indent=1, sort_keys=True)
for name, e in res.items():
    c = e["counters"]
    print("%-16s known %12.0f  FETCH_SIZE %12.0f B (known/it %.3f)  requests %.0f = 32B %.0f + 64B %.0f + 128B %.0f -> %.0f B (known/it %.3f)  ms %.3f" % (
        name, e["known_bytes"], e.get("fetch_size_bytes", 0), e.get("known_over_fetch_size", 0), c.get("TCC_EA0_RDREQ_sum", 0),
        c.get("TCC_EA0_RDREQ_32B_sum", 0), c.get("TCC_EA0_RDREQ_64B_sum", 0), c.get("TCC_EA0_RDREQ_128B_sum", 0),
        e.get("bytes_by_request_size", 0), e.get("known_over_bytes_by_request_size") or 0, e["ms"]))
PY
