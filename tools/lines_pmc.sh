#!/bin/bash
# SQ counters of the detect_lines kernels (one pass, 8 SQ counters): what k_lines_front / k_clahe_blend are bound by
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
O=$R/gpurun_out/r02/pmc_lines_sq
rm -rf $O; mkdir -p $O
cd /tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --kernel-trace -d $O -o pmc --output-format csv -- python3 $R/benchmarks_configs.py lines > $O.log 2>&1 || exit 1
python3 - <<PY
import csv, collections, glob
f = glob.glob("$O/**/pmc_counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("lfa::", "")
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in acc.items():
    if not k.startswith("k_"): continue
    print(k[:40], {c: round(v / 7) for c, v in sorted(d.items())})
PY
