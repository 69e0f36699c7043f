set -o pipefail
mkdir -p gpurun_out/r02y
O=gpurun_out/r02y
export TMPDIR=/tmp
for W in hqc256_tanh hqc192_minsum hqc128_tanh hqc128_mc; do
  EXTRA="--steps 5 --warmup 1 --no-cpu-baseline --pmc off --no-hbm-streaming"
  [ $W = hqc128_mc ] && EXTRA="--trials 1048576 --warmup 1"
  timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/prof_$W -o p --output-format csv -- python3 bench.py --workload $W $EXTRA > $O/$W.log 2>&1; echo "$W rc=$?"
  cp $O/prof_$W/p_kernel_stats.csv $O/${W}_kernel_stats.csv 2>/dev/null
  grep '^{' $O/$W.log | tail -1 > $O/${W}_bench_under_rocprof.json
  head -4 $O/${W}_kernel_stats.csv | cut -c1-150
  rm -rf $O/prof_$W
done
