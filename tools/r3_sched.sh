#!/bin/bash
# configs[2] / configs[3] / photo under the old (2 lanes free) and the new (3 lanes chained) schedule
set -e
cd "$(dirname "$0")/.."
out=gpurun_out/r3_sched.log
: > $out
for cfg in u16 u16real big photo; do
  for mode in "2 0" "3 1" "4 1" "2 1"; do
    set -- $mode
    echo "== $cfg LFA_STREAMS=$1 LFA_CHAIN=$2" >> $out
    LFA_STREAMS=$1 LFA_CHAIN=$2 timeout -k 10 240 python benchmarks_configs.py $cfg >> $out 2>&1
  done
done
