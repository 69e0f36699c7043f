#!/bin/bash
# rocprofv3 --kernel-trace --stats of the bench commands; summaries under $1 (absolute path inside the repo copy)
out=$GRAFT_REPO_ROOT/$1; mkdir -p $out
repo=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
prof() { name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -o p -- python3 $repo/bench.py "$@" --no-cpu-baseline > $out/${name}_bench_under_rocprof.json 2> $out/${name}_rocprof.err || echo "FAILED $name"
  f=$(find /tmp/prof_$name -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/${name}_kernel_stats.csv
  echo "done $name"; }
prof hqc128_minsum --steps 5 --no-hbm-streaming
prof hqc128_mc --workload hqc128_mc --trials 1048576
prof hqc256_tanh --workload hqc256_tanh --steps 3 --no-hbm-streaming
prof qary_config4 --workload qary_config4
prof kyber_sw6_b256 --workload kyber_sw6
