#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
O=$R/gpurun_out/r3_scatter; rm -rf $O; mkdir -p $O
cd /tmp
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O -o pmc --output-format csv -- $R/tools/scatter_probe > $O/probe.txt 2>&1 || { tail -5 $O/probe.txt; exit 1; }
python3 - <<PY
import csv
rows = {}
for r in csv.DictReader(open("$O/pmc_counter_collection.csv")):
    if r["Counter_Name"] == "FETCH_SIZE":
        rows[r["Kernel_Name"].split("(")[0].replace("void ", "")] = float(r["Counter_Value"])
print("kernel, FETCH_SIZE (KiB), piece bytes, bytes in 128-B lines, bytes in 64-B sectors, FETCH_SIZE*1024/line bytes, FETCH_SIZE*1024/sector bytes")
for ln in open("$O/probe.txt"):
    if not ln.startswith("k_scatter"): continue
    t = ln.split()
    name, pb, l128, s64, ms = t[0], int(t[4]), int(t[6]), int(t[8]), float(t[10])
    f = rows.get(name)
    if f is None: print(name, "no counter"); continue
    print(f"{name}, {f:.0f}, {pb}, {l128}, {s64}, {f * 1024 / l128:.3f}, {f * 1024 / s64:.3f}, {ms} ms")
PY
