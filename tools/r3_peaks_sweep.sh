#!/bin/bash
cd "$(dirname "$0")/.."
out=gpurun_out/r3_peaks_sweep.log; : > $out
for lib in "" wpe5 wpe6 wpe8; do for g in 16 32; do for w in 1; do
  echo "== lib=$lib G=$g WAVES=$w" >> $out
  L=; [ -n "$lib" ] && L=image-analysis_amd/variants/liblfa_$lib.so
  LFA_LIB_PATH=$L LFA_PEAKS_G=$g LFA_PEAKS_WAVES=$w timeout -k 10 120 python benchmarks_configs.py rois peaks 2>&1 | grep -o '"ms_per_step": [0-9.]*\|"k_peaks": [0-9.]*\|"ms": [0-9.]*' | tr '\n' ' ' >> $out; echo >> $out
done; done; done
cat $out
