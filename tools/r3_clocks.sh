#!/bin/bash
# Dev aid: GPU clocks and power while the headline step runs (3 chained lanes, then one lane), sampled with rocm-smi.
cd ${GRAFT_REPO_ROOT:-.}
for mode in "--streams 3" "--streams 1"; do
  timeout -k 10 120 python bench.py --no-configs --cpu-sample 0 --no-kernel-timing --steps 1500 $mode > gpurun_out/clk.json 2>/dev/null &
  pid=$!
  sleep 7
  for i in 1 2 3; do
    echo "== $mode sample $i"; timeout 10 rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power|power" | head -8
    sleep 0.7
  done
  wait $pid
  python -c "
import json
d=json.load(open('gpurun_out/clk.json')); print('$mode', d['ms_per_step'])"
done
