#!/bin/bash
# headline step under different lane counts / schedules (dev aid): bash tools/r3_sched_matrix.sh "<streams> <stagger> <lookahead>" ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for cfg in "$@"; do
  set -- $cfg
  timeout -k 10 120 python bench.py --no-configs --cpu-sample 0 --streams $1 --stagger $2 --lookahead $3 > gpurun_out/r3_sched_$1_$2_$3.json 2>/dev/null || exit 1
  python - <<PY
import json
d=json.load(open("gpurun_out/r3_sched_$1_$2_$3.json"))
print("streams $1 stagger $2 lookahead $3:", d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["avg_launch_ms"], flush=True)
PY
done
