#!/bin/bash
# round 3: every BASELINE config's bench line (self-describing: roofline + cpu_baseline), into gpurun_out/r03c/
set -o pipefail
out=gpurun_out/r03c; mkdir -p $out
run() { name=$1; shift; echo "== $name: python bench.py $*"; timeout -k 10 420 python bench.py "$@" > $out/$name.json 2> $out/$name.err; rc=$?; echo "rc=$rc"; tail -c 600 $out/$name.json; echo; return $rc; }
run qary_config4 --workload qary_config4 --steps 20 --warmup 2 &&
run kyber_sw6_b1 --workload kyber_sw6 --batch 1 --steps 10 --warmup 2 &&
run kyber_sw6_b64 --workload kyber_sw6 --batch 64 --steps 5 --warmup 1 &&
run kyber_sw6_b256 --workload kyber_sw6 --batch 256 --steps 5 --warmup 1 &&
run criterion_small --workload criterion_small --steps 200 --warmup 5 &&
run criterion_medium --workload criterion_medium --steps 200 --warmup 5 &&
run hqc128_mc --workload hqc128_mc --trials 1000000 --warmup 1 &&
(python bench.py --gpus 2 --steps 1 > $out/preflight_gpus2.txt 2>&1; echo "preflight rc=$?" >> $out/preflight_gpus2.txt; cat $out/preflight_gpus2.txt)
