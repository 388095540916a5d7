# kernel timeline of the headline bench under a schedule: bash tools/trace_run.sh <stagger>
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 200 rocprofv3 --kernel-trace -d $R/gpurun_out/trace_st$1 -o t --output-format csv -- python3 $R/bench.py --no-configs --cpu-sample 0 --no-kernel-timing --steps 5 --stagger $1 > $R/gpurun_out/trace_st$1.log 2>&1 || exit 1
cd $R
python tools/timeline.py $(find gpurun_out/trace_st$1 -name "*kernel_trace.csv" | head -1) 30 60
