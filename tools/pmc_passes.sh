#!/bin/bash
# rocprofv3 PMC passes over a short bench run (one pass per counter set, each in its own process).
# Run on the GPU box from the repo root:  bash tools/pmc_passes.sh   -> gpurun_out/pmc_<set>/
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
ARGS="--batch 512 --chunk 256 --streams 2 --steps 1 --warmup 1 --cpu-sample 0 --no-kernel-timing"
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/pmc_fetch -o pmc --output-format csv -- python3 $R/bench.py $ARGS > $R/gpurun_out/pmc_fetch.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/pmc_write -o pmc --output-format csv -- python3 $R/bench.py $ARGS > $R/gpurun_out/pmc_write.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY --kernel-trace -d $R/gpurun_out/pmc_sq -o pmc --output-format csv -- python3 $R/bench.py $ARGS > $R/gpurun_out/pmc_sq.log 2>&1
ls $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write $R/gpurun_out/pmc_sq
