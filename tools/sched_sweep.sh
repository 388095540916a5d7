#!/bin/bash
# headline bench under different chunk schedules: lanes (handles + streams), chunk size, free-running / chained passes
# usage (GPU box): bash tools/sched_sweep.sh "<flags 1>" "<flags 2>" ...
for a in "$@"; do
  timeout -k 10 90 python bench.py --no-configs --cpu-sample 0 $a > gpurun_out/sweep.json 2> gpurun_out/sweep.err || exit 1
  python - "$a" <<PY
import json,sys
d=json.loads(open("gpurun_out/sweep.json").read().strip().splitlines()[-1])
print("%-44s step %.3f ms  %.0f strips/s  hist launch %.4f ms frac %.3f (alone %.4f)" % (sys.argv[1], d["ms_per_step"], d["value"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["roofline"]["isolated"]["avg_launch_ms"]), flush=True)
PY
done
