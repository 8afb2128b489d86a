#!/usr/bin/env python3
"""Folds the rocprofv3 CSVs written by tools/collect_profiles.sh into one table per round:
average duration per kernel (kernel trace) and average counter values per kernel launch (PMC).

    python tools/summarize_profiles.py gpurun_out/prof_r01 profiles/r01 [CONFIG]

With CONFIG the per-kernel HBM traffic also goes into profiles/traffic.json, which bench.py reads.
"""
import collections
import csv
import glob
import json
import os
import sys


def find(d, suffix):
    hits = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    return hits[0] if hits else None


def main(src, dst_prefix, config=None):
    rows = {}
    stats = find(os.path.join(src, "trace"), "kernel_stats.csv")
    if stats:
        for r in csv.DictReader(open(stats)):
            rows[r["Name"]] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                               "min_us": float(r["MinNs"]) / 1e3, "max_us": float(r["MaxNs"]) / 1e3}
        with open(dst_prefix + "_kernel_stats.csv", "w") as f:
            f.write(open(stats).read())
    for sub in sorted(glob.glob(os.path.join(src, "pmc_*"))):
        if not os.path.isdir(sub):
            continue
        cc = find(sub, "counter_collection.csv")
        if not cc:
            continue
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        meta = {}
        for r in csv.DictReader(open(cc)):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta[r["Kernel_Name"]] = {"vgpr": int(r["VGPR_Count"]), "lds": int(r["LDS_Block_Size"]),
                                      "scratch": int(r["Scratch_Size"]), "wg": int(r["Workgroup_Size"]),
                                      "grid": int(r["Grid_Size"])}
        for k, counters in acc.items():
            e = rows.setdefault(k, {})
            e.update(meta[k])
            for name, vals in counters.items():
                e[name] = sum(vals) / len(vals)
    names = sorted(rows, key=lambda n: -rows[n].get("avg_us", 0) * rows[n].get("calls", 1))
    cols = sorted({c for r in rows.values() for c in r})
    with open(dst_prefix + "_pmc.csv", "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel"] + cols)
        for n in names:
            w.writerow([n] + [rows[n].get(c, "") for c in cols])
    # HBM traffic per launch for the benchmark's roofline line.  FETCH_SIZE / WRITE_SIZE are in KiB;
    # on gfx950 FETCH_SIZE counts every read request as 64 bytes while the requests are 128-byte ones, for wide
    # streaming reads and for isolated 4-byte gathers alike (tools/micro/fetch_calib.hip, profiles/r02_fetch_calib.json),
    # so reads are doubled.  When the pass with the size-resolved request counters was collected, the read bytes are
    # taken from it instead (32 / 64 / 128 bytes per request of each size: no correction needed).
    traffic = {}
    for n, r in rows.items():
        if "FETCH_SIZE" in r and "WRITE_SIZE" in r:
            short = n.split("(")[0].replace("void ", "").replace("rnd::", "")
            read = 2.0 * r["FETCH_SIZE"] * 1024.0
            source = "FETCH_SIZE x 2"
            if "TCC_EA0_RDREQ_128B_sum" in r:
                read = 32.0 * r.get("TCC_EA0_RDREQ_32B_sum", 0.0) + 64.0 * r.get("TCC_EA0_RDREQ_64B_sum", 0.0) + 128.0 * r["TCC_EA0_RDREQ_128B_sum"]
                source = "TCC_EA0_RDREQ by request size"
            traffic[short] = {"read_bytes": read, "read_bytes_source": source, "read_bytes_fetch_size_x2": 2.0 * r["FETCH_SIZE"] * 1024.0,
                              "write_bytes": r["WRITE_SIZE"] * 1024.0, "avg_us": r.get("avg_us"), "valu_insts": r.get("SQ_INSTS_VALU")}
    with open(dst_prefix + "_traffic.json", "w") as f:
        json.dump(traffic, f, indent=1, sort_keys=True)
    print("wrote", dst_prefix + "_pmc.csv", dst_prefix + "_traffic.json")
    if config is not None:
        # the table bench.py reads for roofline.traffic: one entry per benchmark configuration
        path = os.path.join(os.path.dirname(dst_prefix) or ".", "traffic.json")
        try:
            table = json.load(open(path))
        except (OSError, ValueError):
            table = {}
        table["config%d" % config] = {k: v for k, v in traffic.items() if k.startswith("k_")}
        # which build the pass belongs to: bench.py reports the counters only for the same kernel sources
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        table.setdefault("collected", {})["config%d" % config] = {"tag": os.path.basename(dst_prefix), "kernel_source_id": bench.kernel_source_id()}
        with open(path, "w") as f:
            json.dump(table, f, indent=1, sort_keys=True)
        print("updated", path)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else None)
