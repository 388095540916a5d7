#!/usr/bin/env python3
"""profiles/r03_<cfg>_kernel_stats.csv (rocprofv3 --stats durations) x profiles/r03_pmc_hbm_traffic.csv (PMC bytes per launch)
-> profiles/r03_stage_hbm.csv: achieved HBM GB/s of every kernel against the 8 TB/s peak, per configuration.
Durations are averages over the launches of the profiled run (the headline run: the timed region's chained schedule, where the
whole-image passes run one at a time and the small kernels beside them; `min_ns` is the kernel with the chip to itself)."""
import os

P = "profiles"
PEAK = 8000.0
traffic = {}
for ln in open(f"{P}/r03_pmc_hbm_traffic.csv"):
    if ln.startswith(("#", "config,")):
        continue
    head, launches, fetch, write, nbytes = ln.strip().rsplit(",", 4)          # (Gaussian kernel names hold commas)
    cfg, _cmd, kernel = head.split(",", 2)
    traffic[(cfg, kernel)] = int(nbytes)
rows = []
for cfg in ("bench", "cfg2", "photo", "cfg4", "gauss", "lines"):
    path = f"{P}/r03_{cfg}_kernel_stats.csv"
    if not os.path.exists(path):
        continue
    for ln in open(path):
        if ln.startswith(("#", "kernel,")):
            continue
        kernel, calls, _tot, avg, _pct, mn, _mx, _sd = ln.strip().rsplit(",", 7)
        b = traffic.get((cfg, kernel))
        if not b:
            continue
        avg, mn = float(avg), float(mn)
        rows.append((cfg, '"' + kernel + '"' if "," in kernel else kernel, int(calls), b, avg, b / avg, b / avg / PEAK, mn, b / mn, b / mn / PEAK))
with open(f"{P}/r03_stage_hbm.csv", "w") as fo:
    fo.write("# HBM bytes per launch from the PMC passes (2*FETCH_SIZE + WRITE_SIZE, KiB) / launch duration from rocprofv3 --stats; GB/s against 8000 GB/s\n")
    fo.write("# 2*FETCH_SIZE = bytes of the 128-byte lines fetched: calibrated for streaming reads (MI355X_MICROARCH.md) and for scattered 16..64-byte pieces (profiles/r03_scatter_probe.csv), i.e. for every row\n")
    fo.write("# durations are those of the PIPELINED run: a tail kernel's launch lasts from its dispatch to its last block, and its blocks wait for CUs while another chunk's image pass holds them (the empty k_mask of a settled handle: 290 us; k_profile behind its LDS gate: the rest of a histogram pass) -- for the tail kernels the *_best columns and the one-stream times in the bench line's kernel_ms_per_step are the kernel, the averages are the schedule (DESIGN.md, 'Where the tail kernels run')\n")
    fo.write("config,kernel,launches,hbm_bytes_per_launch,average_ns,GBps_average,frac_of_peak_average,min_ns,GBps_best,frac_of_peak_best\n")
    for r in rows:
        fo.write(f"{r[0]},{r[1]},{r[2]},{r[3]},{r[4]:.0f},{r[5]:.1f},{r[6]:.3f},{r[7]:.0f},{r[8]:.1f},{r[9]:.3f}\n")
print(open(f"{P}/r03_stage_hbm.csv").read())
