#!/bin/bash
# Round-3 evidence run on the GPU box (from the repo root): rocprofv3 --kernel-trace --stats of the headline bench and of
# the secondary configurations, then FETCH_SIZE / WRITE_SIZE passes (each counter set in a run of its own) for each.
# Everything lands in gpurun_out/r03/; tools/round3_summarize.py turns it into the files kept under profiles/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
O=$R/gpurun_out/r03
mkdir -p $O
cd /tmp
run_stats() {   # name, program args...
  name=$1; shift
  timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $O/stats_$name -o s --output-format csv -- python3 "$@" > $O/stats_$name.log 2>&1 || return 1
}
run_pmc() {     # name, counter, program args...
  name=$1; ctr=$2; shift; shift
  timeout -s KILL 300 rocprofv3 --pmc $ctr --kernel-trace -d $O/pmc_${name}_$ctr -o pmc --output-format csv -- python3 "$@" > $O/pmc_${name}_$ctr.log 2>&1 || return 1
}
HEAD="$R/bench.py --cpu-sample 0 --no-configs --no-kernel-timing"
run_stats bench $R/bench.py --cpu-sample 0 --no-configs && grep "^{" $O/stats_bench.log | tail -1 > $O/bench_under_rocprof.json && echo stats bench ok &&
run_stats cfg2 $R/benchmarks_configs.py u16 && echo stats cfg2 ok &&
run_stats cfg3 $R/benchmarks_configs.py big && echo stats cfg3 ok &&
run_stats cfg4 $R/benchmarks_configs.py rois && echo stats cfg4 ok &&
run_stats photo $R/benchmarks_configs.py photo && echo stats photo ok &&
run_stats gauss $R/benchmarks_configs.py gauss && echo stats gauss ok &&
run_stats lines $R/benchmarks_configs.py lines && echo stats lines ok &&
run_pmc bench FETCH_SIZE $HEAD --batch 1024 --steps 1 --warmup 3 && run_pmc bench WRITE_SIZE $HEAD --batch 1024 --steps 1 --warmup 3 && echo pmc bench ok &&
run_pmc cfg2 FETCH_SIZE $R/benchmarks_configs.py u16 && run_pmc cfg2 WRITE_SIZE $R/benchmarks_configs.py u16 && echo pmc cfg2 ok &&
run_pmc photo FETCH_SIZE $R/benchmarks_configs.py photo && run_pmc photo WRITE_SIZE $R/benchmarks_configs.py photo && echo pmc photo ok &&
run_pmc cfg4 FETCH_SIZE $R/benchmarks_configs.py rois && run_pmc cfg4 WRITE_SIZE $R/benchmarks_configs.py rois && echo pmc cfg4 ok &&
run_pmc gauss FETCH_SIZE $R/benchmarks_configs.py gauss && run_pmc gauss WRITE_SIZE $R/benchmarks_configs.py gauss && echo pmc gauss ok &&
run_pmc lines FETCH_SIZE $R/benchmarks_configs.py lines && run_pmc lines WRITE_SIZE $R/benchmarks_configs.py lines && echo pmc lines ok
find $O -name "*.csv" | head -40
