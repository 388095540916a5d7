#!/bin/bash
# A/B of library variants (tools/build_variant.py) on the headline bench: ms per step, isolated time of the dominant kernel
# usage (on the GPU box): bash tools/ab.sh "<bench flags>" name1 name2 ...
flags="$1"; shift
for v in "$@"; do
  LFA_LIB_PATH=image-analysis_amd/variants/liblfa_$v.so timeout -k 10 90 python bench.py --no-configs --cpu-sample 0 $flags > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err || exit 1
  python - "$v" <<PY
import json,sys
d=json.loads(open("gpurun_out/ab_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
k=d["kernel_ms_per_step"]
print(sys.argv[1], "step", d["ms_per_step"], "hist shared", d["roofline"]["avg_launch_ms"], "isolated", d["roofline"]["isolated"]["avg_launch_ms"], "one-stream per step:", {n:k[n] for n in list(k)[:4]}, flush=True)
PY
done
