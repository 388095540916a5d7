# rocprofv3 --kernel-trace --stats of the default headline command (no CPU baseline / secondary configurations under the profiler)
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
O=$R/gpurun_out/r02
mkdir -p $O; rm -rf $O/stats_bench
cd /tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $O/stats_bench -o s --output-format csv -- python3 $R/bench.py --cpu-sample 0 --no-configs > $O/stats_bench.log 2>&1 || exit 1
grep '^{' $O/stats_bench.log > $O/stats_bench.json
echo stats bench ok
