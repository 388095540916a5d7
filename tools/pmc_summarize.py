#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc passes of tools/pmc_passes.sh (gpurun_out/pmc_*/pmc_counter_collection.csv).

    python tools/pmc_summarize.py gpurun_out profiles/r01_pmc_hbm_traffic.csv profiles/traffic.json 1536x2048x3 256

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: rocprofv3 reports KiB, and on gfx950 FETCH_SIZE counts half
of a wide coalesced read (MI355X_MICROARCH.md, section HBM)."""
import collections
import csv
import json
import re
import sys


def per_kernel(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        m = re.search(r"lfa::(\w+)", r["Kernel_Name"])
        if not m:
            continue
        acc[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}


def main():
    src, out_csv, out_json, shape, chunk = sys.argv[1:6]
    fetch = per_kernel(f"{src}/pmc_fetch/pmc_counter_collection.csv")
    write = per_kernel(f"{src}/pmc_write/pmc_counter_collection.csv")
    rows = []
    for k in fetch:
        f = fetch[k].get("FETCH_SIZE", 0.0)
        w = write.get(k, {}).get("WRITE_SIZE", 0.0)
        rows.append((k, f, w, int((2 * f + w) * 1024)))
    rows.sort(key=lambda r: -r[3])
    with open(out_csv, "w") as fo:
        fo.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/pmc_passes.sh): bench.py --batch 512 --chunk "
                 f"{chunk} --streams 2 --steps 1 --warmup 1, {shape} uint8\n")
        fo.write("# raw counters are KiB per launch (mean over the launches of one chunk); hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: "
                 "on gfx950 FETCH_SIZE reports half of a wide coalesced read (MI355X_MICROARCH.md, HBM)\n")
        fo.write("kernel,FETCH_SIZE_KiB,WRITE_SIZE_KiB,hbm_bytes_per_launch,hbm_bytes_per_image\n")
        for k, f, w, b in rows:
            fo.write(f"lfa::{k},{f:.1f},{w:.1f},{b},{b // int(chunk)}\n")
    tj = {f"{k}:{shape}:chunk{chunk}": b for k, f, w, b in rows[:6]}
    json.dump(tj, open(out_json, "w"), indent=1)
    try:
        sq = per_kernel(f"{src}/pmc_sq/pmc_counter_collection.csv")
    except FileNotFoundError:
        return
    names = ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_LDS",
             "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT"]
    with open(out_csv.replace("hbm_traffic", "sq_lds"), "w") as fo:
        fo.write("# rocprofv3 --pmc SQ_* pass of the same run; per launch; *_frac are fractions of SQ_WAVE_CYCLES\n")
        fo.write("kernel," + ",".join(names) + ",wait_any_frac,lds_conflict_frac_of_lds_active\n")
        for k, d in sorted(sq.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
            wc = d.get("SQ_WAVE_CYCLES", 0) or 1
            la = d.get("SQ_LDS_IDX_ACTIVE", 0) or 1
            fo.write(f"lfa::{k}," + ",".join(f"{d.get(n, 0):.0f}" for n in names) +
                     f",{d.get('SQ_WAIT_ANY', 0) / wc:.3f},{d.get('SQ_LDS_BANK_CONFLICT', 0) / la:.3f}\n")


if __name__ == "__main__":
    main()
