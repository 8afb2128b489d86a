import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows = [r for r in rows if "k_tile" in r["Kernel_Name"] or "k_setup" in r["Kernel_Name"] or "k_vertex" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[-60]["Start_Timestamp"])
for r in rows[-60:]:
    n = r["Kernel_Name"]
    n = "tile" if "k_tile" in n else "setup" if "k_setup" in n else "vertex"
    print("%-7s q%-3s start %8.1f end %8.1f dur %6.1f" % (n, r.get("Queue_Id", "?"), (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
