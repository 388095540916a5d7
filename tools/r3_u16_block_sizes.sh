#!/bin/bash
# configs[2] (1024 frames) under k_hist_u16 block sizes (dev aid)
for t in 1024 768 512 256; do for p in 64 128; do
LFA_U16_THREADS=$t LFA_U16_PPT=$p timeout -k 10 120 python benchmarks_configs.py u16 --json > gpurun_out/u16_$t_$p.json 2>/dev/null || exit 1
python - <<PY
import json
d=[json.loads(l) for l in open("gpurun_out/u16_$t_$p.json") if l.startswith("{")][-1]
c=list(d.values())[0]
print("threads $t ppt $p", c["ms_per_step"], c["dominant"]["ms_per_launch"], {k:v for k,v in list(c["kernel_ms_one_stream"].items())[:3]}, flush=True)
PY
done; done
