#!/bin/bash
# round 3: rocprofv3 --kernel-trace --stats of every bench workload -> gpurun_out/r03d/ (summaries copied to profiles/r03/)
set -o pipefail
O=gpurun_out/r03d; mkdir -p $O
export TMPDIR=/tmp
prof() { W=$1; shift
  timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $O/prof_$W -o p --output-format csv -- python3 bench.py "$@" > $O/$W.log 2>&1; rc=$?; echo "$W rc=$rc"
  cp $O/prof_$W/p_kernel_stats.csv $O/${W}_kernel_stats.csv 2>/dev/null
  grep '^{' $O/$W.log | tail -1 > $O/${W}_bench_under_rocprof.json
  head -4 $O/${W}_kernel_stats.csv | cut -c1-160
  rm -rf $O/prof_$W
  return $rc; }
prof hqc128_minsum --workload hqc128_minsum --steps 5 --warmup 1 --no-cpu-baseline --pmc off --no-hbm-streaming &&
prof qary_config4 --workload qary_config4 --steps 20 --warmup 2 --no-cpu-baseline &&
prof kyber_sw6_b256 --workload kyber_sw6 --batch 256 --steps 5 --warmup 1 --no-cpu-baseline &&
prof kyber_sw6_b1 --workload kyber_sw6 --batch 1 --steps 20 --warmup 2 --no-cpu-baseline &&
prof criterion_small --workload criterion_small --steps 200 --warmup 5 --no-cpu-baseline &&
prof criterion_medium --workload criterion_medium --steps 200 --warmup 5 --no-cpu-baseline &&
prof hqc128_mc --workload hqc128_mc --trials 1048576 --warmup 1 --no-cpu-baseline --parity-rows 0 &&
(timeout -k 10 300 python3 bench.py > $O/bench_default_run.log 2>&1; echo "default rc=$?"; tail -c 400 $O/bench_default_run.log)
