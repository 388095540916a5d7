#!/bin/bash
# SQ counters of one benchmarks_configs.py configuration (dev aid):  bash tools/pmc_kernel.sh <config> ; then
# python tools/pmc_kernel_sum.py gpurun_out/pmc_k <kernel name substring>
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU --kernel-trace -d $R/gpurun_out/pmc_k -o pmc --output-format csv -- python3 $R/benchmarks_configs.py $1 > $R/gpurun_out/pmc_k.log 2>&1 || exit 1
