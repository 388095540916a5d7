#!/bin/bash
# Dev aid: full bench.py (secondary configurations included, no CPU baseline) for two builds of the library, alternating.
# usage: tools/ab_full.sh <a.so> <b.so> [repeats]
for rep in $(seq 1 ${3:-1}); do for lib in $1 $2; do
  LFA_LIB_PATH=$lib timeout -k 10 300 python bench.py --cpu-sample 0 > gpurun_out/abf.json 2>/dev/null || exit 1
  python -c "
import json
d=json.load(open('gpurun_out/abf.json')); print('$lib'.split('liblfa_')[-1], d['ms_per_step'], d['roofline']['frac'], {k:(v.get('ms_per_step') if isinstance(v,dict) else v) for k,v in d.get('configs',{}).items()})"
done; done
