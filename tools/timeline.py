"""Print the kernel timeline of the last steps of a rocprofv3 --kernel-trace CSV (one line per launch: start, end, duration in
microseconds, queue, kernel) and the time during which none of the three image passes was running.
usage: python tools/timeline.py <kernel_trace.csv> [first_minmax_from_the_end=8] [rows=80]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 8
count = int(sys.argv[3]) if len(sys.argv) > 3 else 80
r = sorted((int(x["Start_Timestamp"]), int(x["End_Timestamp"]),
            x["Kernel_Name"].split("(")[0].replace("void ", "").replace("lfa::", ""), x["Queue_Id"]) for x in rows)
names = [x[2] for x in r]
first = [i for i, n in enumerate(names) if n.startswith(("k_minmax_rgb", "k_hist_u16"))][-back]
t0 = r[first][0]
big = ("k_minmax", "k_hist", "k_mask")
busy_until, idle = t0, 0
for s, e, n, q in r[first:first + count]:
    print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  q{q} {n[:30]}")
    if n.startswith(big):
        if s > busy_until:
            idle += s - busy_until
        busy_until = max(busy_until, e)
print(f"no image pass running: {idle / 1e3:.1f} us of {(busy_until - t0) / 1e3:.1f} us")
