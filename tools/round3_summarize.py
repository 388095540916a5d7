#!/usr/bin/env python3
"""gpurun_out/r03/ (tools/round3_profile.sh) -> the summaries kept under profiles/:
  r03_<cfg>_kernel_stats.csv   rocprofv3 --kernel-trace --stats of each configuration (lfa:: kernels only)
  r03_pmc_hbm_traffic.csv      HBM bytes per launch and kernel from the FETCH_SIZE / WRITE_SIZE passes
  traffic.json                 the headline kernels' figures for bench.py's roofline.traffic, with their source
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: rocprofv3 reports KiB, and on gfx950 FETCH_SIZE counts every 128-byte
line the L2 requests as 64 bytes -- for wide coalesced reads (MI355X_MICROARCH.md, section HBM) and, measured this round with
tools/scatter_probe.hip (profiles/r03_scatter_probe.csv), for scattered pieces of 16..64 bytes alike: 2 * FETCH_SIZE is the
bytes of the 128-byte lines touched, whatever part of a line the kernel uses."""
import collections
import csv
import json
import os
import re
import sys

SRC = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r03"
DST = sys.argv[2] if len(sys.argv) > 2 else "profiles"
CMD = {"bench": "bench.py --cpu-sample 0 --no-configs (stats; the default schedule only, its JSON line in r03_bench_under_rocprof.json) / ... --no-kernel-timing --batch 1024 --steps 1 --warmup 3 (pmc)",
       "cfg2": "benchmarks_configs.py u16", "cfg3": "benchmarks_configs.py big", "cfg4": "benchmarks_configs.py rois", "photo": "benchmarks_configs.py photo",
       "gauss": "benchmarks_configs.py gauss", "lines": "benchmarks_configs.py lines"}


def short(name):
    m = re.search(r"lfa::(\w+)", name)
    if not m:
        return None
    g = re.search(r"lfa::k_gaussian(?:_f32)?<(unsigned char|unsigned short)(?:, (double|float))?, (\d)", name)
    if g:                                   # the Gaussian kernels run on three shapes: keep pixel type, output type and channels
        return m.group(1) + "<" + ("u8" if g.group(1) == "unsigned char" else "u16") + "," + \
            ("f32" if (g.group(2) == "float" or "_f32" in m.group(1)) else "f64") + ",C" + g.group(3) + ">"
    return m.group(1)


for cfg in CMD:
    p = f"{SRC}/stats_{cfg}/s_kernel_stats.csv"
    if not os.path.exists(p):
        continue
    rows = list(csv.DictReader(open(p)))
    with open(f"{DST}/r03_{cfg}_kernel_stats.csv", "w") as fo:
        fo.write(f"# rocprofv3 --kernel-trace --stats -- python3 {CMD[cfg]}   (durations in ns; lfa:: kernels only, template arguments cut)\n")
        fo.write("kernel,calls,total_ns,average_ns,percent,min_ns,max_ns,stddev_ns\n")
        for r in rows:
            k = short(r["Name"])
            if k:
                fo.write(f"lfa::{k},{r['Calls']},{r['TotalDurationNs']},{float(r['AverageNs']):.0f},{float(r['Percentage']):.2f},"
                         f"{r['MinNs']},{r['MaxNs']},{float(r['StdDev']):.0f}\n")


def per_kernel(path):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        if k:
            acc[k].append(float(r["Counter_Value"]))
    # median over the launches: the first calls of a handle take other paths (no plane yet, then the class plane) than the
    # settled ones (predicted-threshold bit plane), and the settled launches are the majority of a run with three warm-up steps
    return {k: (sorted(v)[len(v) // 2], len(v)) for k, v in acc.items()}


traffic = {"_source": "profiles/r03_pmc_hbm_traffic.csv (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate runs of "
                      "bench.py --batch 1024 --chunk 256 --steps 1 --warmup 3; tools/round3_profile.sh; median over a kernel's launches)"}
with open(f"{DST}/r03_pmc_hbm_traffic.csv", "w") as fo:
    fo.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, one counter per run (tools/round3_profile.sh); median over the launches of a kernel\n")
    fo.write("# hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 (KiB units; gfx950 FETCH_SIZE tallies each 128-byte line request at 64 bytes: calibrated for\n")
    fo.write("# coalesced streaming reads by MI355X_MICROARCH.md and for scattered 16..64-byte pieces by tools/scatter_probe.hip, profiles/r03_scatter_probe.csv -- every row below)\n")
    fo.write("config,command,kernel,launches,FETCH_SIZE_KiB,WRITE_SIZE_KiB,hbm_bytes_per_launch\n")
    for cfg in CMD:
        pf, pw = f"{SRC}/pmc_{cfg}_FETCH_SIZE/pmc_counter_collection.csv", f"{SRC}/pmc_{cfg}_WRITE_SIZE/pmc_counter_collection.csv"
        if not (os.path.exists(pf) and os.path.exists(pw)):
            continue
        f, w = per_kernel(pf), per_kernel(pw)
        rows = sorted(((k, f[k][0], w.get(k, (0.0, 0))[0], f[k][1]) for k in f), key=lambda r: -(2 * r[1] + r[2]))
        for k, fv, wv, n in rows:
            b = int((2 * fv + wv) * 1024)
            fo.write(f"{cfg},{CMD[cfg].split(' (')[0]},lfa::{k},{n},{fv:.1f},{wv:.1f},{b}\n")
            if cfg == "bench":
                traffic[f"{k}:1536x2048x3:chunk256"] = b
json.dump(traffic, open(f"{DST}/traffic.json", "w"), indent=1)
print(open(f"{DST}/r03_pmc_hbm_traffic.csv").read())
