#!/bin/bash
# Dev aid: headline step time over blocks per image of the two image passes (LFA_MM_BPI x LFA_HIST_BPI; 0 = the library's own choice).
# usage: [BENCH_EXTRA=--no-kernel-timing] tools/r3_bpi_matrix.sh <lib.so> "<mm values>" "<hist values>"   (the timing events around the dominant kernel shift the optimum)
LIB=$1
for m in $2; do for h in $3; do
  LFA_MM_BPI=$m LFA_HIST_BPI=$h LFA_LIB_PATH=$LIB timeout -k 10 90 python bench.py --no-configs --cpu-sample 0 $BENCH_EXTRA > gpurun_out/bpi_${m}_${h}.json 2>/dev/null || exit 1
  python -c "
import json
d=json.load(open('gpurun_out/bpi_${m}_${h}.json')); print('mm $m hist $h', d['ms_per_step'])"
done; done
