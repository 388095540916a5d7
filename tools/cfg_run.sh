for c in u16 rois u16real lines; do timeout -k 10 280 python benchmarks_configs.py $c --json --steps 10 > gpurun_out/cfg_$c.json 2> gpurun_out/cfg_$c.err || { echo FAIL $c; tail -5 gpurun_out/cfg_$c.err; exit 1; }; python - $c <<PY
import json,sys
d=json.loads(open("gpurun_out/cfg_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
for k,v in d.items(): print(k, {q:v[q] for q in v if q in ("config","ms_per_step","frames_per_s","rois_per_s","e2e_frac_hbm","dominant")})
PY
done
