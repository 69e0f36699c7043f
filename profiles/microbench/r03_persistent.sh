#!/bin/bash
# round 3: the one-launch single-decode microbenchmark + kernel trace of the product's 100-iteration single decode
export TMPDIR=/tmp; O=gpurun_out/r03e; mkdir -p $O
timeout -k 10 120 ./profiles/microbench/persistent_bp > $O/persistent_bp.log 2>&1; echo "rc=$?"; cat $O/persistent_bp.log
for M in product_sum min_sum; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/p_$M -o p --output-format csv -- python3 profiles/microbench/single_decode_profile.py $M > $O/single_$M.log 2>&1
  cp $O/p_$M/p_kernel_stats.csv $O/single_decode_${M}_kernel_stats.csv; rm -rf $O/p_$M; tail -1 $O/single_$M.log
done
