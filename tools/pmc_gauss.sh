#!/bin/bash
# SQ counter passes over the Gaussian stage alone (dev aid): where do the wave cycles go?
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $R/gpurun_out/pmc_gauss_a -o pmc --output-format csv -- python3 $R/benchmarks_configs.py gauss > $R/gpurun_out/pmc_gauss_a.log 2>&1 || exit 1
ls $R/gpurun_out/pmc_gauss_a
