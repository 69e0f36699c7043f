#!/bin/bash
# every bench workload once with the shipped library; one JSON line each under $1
out=$1; mkdir -p $out
run() { name=$1; shift; timeout -k 10 420 python bench.py "$@" > $out/bench_$name.json 2> $out/bench_$name.err || echo "FAILED $name rc=$?"; tail -c 300 $out/bench_$name.err | grep -v "amdgpu.ids" | tail -2; echo "done $name"; }
run hqc128_minsum --pmc-save $out/pmc_traffic_hqc128_minsum.json
run hqc192_minsum --workload hqc192_minsum --pmc-save $out/pmc_traffic_hqc192_minsum.json
run hqc128_tanh --workload hqc128_tanh --pmc-save $out/pmc_traffic_hqc128_tanh.json
run hqc256_tanh --workload hqc256_tanh --pmc-save $out/pmc_traffic_hqc256_tanh.json
run hqc128_mc --workload hqc128_mc --trials 1048576
run qary_config4 --workload qary_config4 --pmc-save $out/sq_counters_qary_config4_b1024.json
run kyber_sw6_b256 --workload kyber_sw6 --pmc-save $out/sq_counters_kyber_sw6_b256.json
run kyber_sw6_b64 --workload kyber_sw6 --batch 64 --pmc-save $out/sq_counters_kyber_sw6_b64.json
run kyber_sw6_b1 --workload kyber_sw6 --batch 1 --pmc-save $out/sq_counters_kyber_sw6_b1.json
run criterion_small --workload criterion_small
run criterion_medium --workload criterion_medium
