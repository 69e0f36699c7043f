#!/bin/bash
# the Kyber workload's bench lines (live SQ pass saved as the committed fallback) and its rocprofv3 kernel stats; summaries under $1
out=$GRAFT_REPO_ROOT/$1; mkdir -p $out
repo=$GRAFT_REPO_ROOT
run() { name=$1; shift; timeout -k 10 420 python bench.py "$@" > $out/bench_$name.json 2> $out/bench_$name.err || echo "FAILED $name rc=$?"; echo "done $name"; }
run kyber_sw6_b256 --workload kyber_sw6 --pmc-save $out/sq_counters_kyber_sw6_b256.json
run kyber_sw6_b64 --workload kyber_sw6 --batch 64 --pmc-save $out/sq_counters_kyber_sw6_b64.json
run kyber_sw6_b1 --workload kyber_sw6 --batch 1 --pmc-save $out/sq_counters_kyber_sw6_b1.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kyber -o p -- python3 $repo/bench.py --workload kyber_sw6 --no-cpu-baseline > $out/kyber_sw6_b256_bench_under_rocprof.json 2> $out/kyber_sw6_b256_rocprof.err || echo "FAILED rocprof"
f=$(find /tmp/prof_kyber -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/kyber_sw6_b256_kernel_stats.csv
echo "done rocprof"
