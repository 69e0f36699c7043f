#!/bin/bash
# config 4's bench line (live SQ pass saved as the committed fallback) and its rocprofv3 kernel stats; summaries under $1
out=$GRAFT_REPO_ROOT/$1; mkdir -p $out
repo=$GRAFT_REPO_ROOT
timeout -k 10 420 python bench.py --workload qary_config4 --pmc-save $out/sq_counters_qary_config4_b1024.json > $out/bench_qary_config4.json 2> $out/bench_qary_config4.err || echo "FAILED bench"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_q4 -o p -- python3 $repo/bench.py --workload qary_config4 --no-cpu-baseline > $out/qary_config4_bench_under_rocprof.json 2> $out/qary_config4_rocprof.err || echo "FAILED rocprof"
f=$(find /tmp/prof_q4 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/qary_config4_kernel_stats.csv
echo "done"
