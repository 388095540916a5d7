#!/bin/bash
# One rocprofv3 --pmc pass over a benchmarks_configs.py configuration:  tools/r3_pmc.sh <config> <tag> <counter> [<counter> ...]
# (at most 8 SQ_, 4 TCC_, 2 GRBM_ counters per pass); summary with tools/pmc_kernel_sum.py gpurun_out/<tag> <kernel substring>
R=${GRAFT_REPO_ROOT:-$(pwd)}
cfg=$1; tag=$2; shift 2
export TMPDIR=/tmp
O=$R/gpurun_out/$tag
rm -rf $O; mkdir -p $O
cd /tmp
timeout -s KILL 240 rocprofv3 --pmc "$@" --kernel-trace -d $O -o pmc --output-format csv -- python3 $R/benchmarks_configs.py $cfg > $O.log 2>&1 || { tail -5 $O.log; exit 1; }
