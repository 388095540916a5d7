#!/bin/bash
# One rocprofv3 --pmc pass over the headline bench (kernels serialised by the profiler): tools/r3_pmc_bench.sh <tag> "<bench flags>" <counter> ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; flags=$2; shift 2
export TMPDIR=/tmp
O=$R/gpurun_out/$tag
rm -rf $O; mkdir -p $O
cd /tmp
timeout -s KILL 240 rocprofv3 --pmc "$@" --kernel-trace -d $O -o pmc --output-format csv -- python3 $R/bench.py --no-configs --cpu-sample 0 --no-kernel-timing --steps 2 --warmup 2 $flags > $O.log 2>&1 || { tail -5 $O.log; exit 1; }
cd $R
python tools/pmc_kernel_sum.py gpurun_out/$tag k_hist
