#!/bin/bash
# End-of-round evidence run on the GPU box (from the repo root): parity tests, the default bench, the same bench under
# rocprofv3 --kernel-trace --stats, the PMC passes, the secondary configurations.  Everything lands in gpurun_out/.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd $R
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/round_tests.log 2>&1
tail -2 gpurun_out/round_tests.log
timeout -k 10 300 python bench.py > gpurun_out/round_bench.log 2>&1
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/round_prof -o r --output-format csv -- python3 $R/bench.py --cpu-sample 0 --no-configs > $R/gpurun_out/round_prof_bench.log 2>&1
cd $R
bash tools/pmc_passes.sh > gpurun_out/round_pmc.log 2>&1
# one process per configuration: measured in one process after another configuration, configs[3] came out 14 % lower
# (allocator / placement state left behind), alone it repeats within 0.3 %
: > gpurun_out/round_secondary.log
for c in u16 u16real big photo rois gauss peaks stream; do
  timeout -k 10 200 python benchmarks_configs.py $c >> gpurun_out/round_secondary.log 2>&1
done
echo done
