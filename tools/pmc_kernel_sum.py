import csv, collections, sys
d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc_k"
pat = sys.argv[2] if len(sys.argv) > 2 else ""
dur = collections.defaultdict(list)
for r in csv.DictReader(open(f"{d}/pmc_kernel_trace.csv")):
    if pat in r["Kernel_Name"]:
        dur[r["Kernel_Name"][:70]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f"{d}/pmc_counter_collection.csv")):
    if pat in r["Kernel_Name"]:
        agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    us = sorted(dur[k])[len(dur[k]) // 2]
    print(k, f"median {us:.1f} us, {len(dur[k])} launches")
    for c, vals in sorted(v.items()):
        m = sorted(vals)[len(vals) // 2]
        print(f"    {c:24s} {m:12.4g}   per us {m / us:10.4g}")
