#!/bin/bash
# kernel timeline of configs[2] (benchmarks_configs.py u16, 1024 frames) under the default pipeline schedule
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
rm -rf $R/gpurun_out/trace_u16
timeout -s KILL 200 rocprofv3 --kernel-trace -d $R/gpurun_out/trace_u16 -o t --output-format csv -- python3 $R/benchmarks_configs.py u16 > $R/gpurun_out/trace_u16.log 2>&1 || exit 1
cd $R
python tools/timeline.py $(find gpurun_out/trace_u16 -name "*kernel_trace.csv" | head -1) 14 110
