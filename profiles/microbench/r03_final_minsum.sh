#!/bin/bash
# round 3, end of round (after the min-sum record form): the HQC workloads' bench lines, the rocprofv3 kernel stats of
# the headline workload and the default bench run (cpu_baseline, live PMC) -> gpurun_out/r03ar/
export TMPDIR=/tmp; O=gpurun_out/r03ar; mkdir -p $O; : > $O/all_workloads.log
for W in hqc128_minsum hqc128_tanh hqc192_minsum hqc256_tanh; do
  timeout -k 10 400 python bench.py --workload $W --steps 5 --warmup 1 --no-cpu-baseline > $O/$W.json 2> $O/$W.err; echo "$W rc=$?"
  grep '^{' $O/$W.json | tail -1 >> $O/all_workloads.log
done
python - <<'PY'
import json
for l in open("gpurun_out/r03ar/all_workloads.log"):
    d=json.loads(l); r=d["roofline"]
    print(d["config"]["workload"][:60].ljust(62), "%.4g upd/s"%d["value"], "%.2f ms"%d["ms_per_step"], "whole-job frac %.3f"%(d["whole_job_algorithmic_GBps"]/8000), "pair %.3f"%r["frac"], r["kernel"], "dominant alone", r["dominant"]["frac"] and round(r["dominant"]["frac"],3), "traffic x%.3f"%(r["traffic"]/r["algorithmic_bytes_per_launch"]) if r["traffic"] else "", d["parity_ok"])
PY
W=hqc128_minsum
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $O/prof_$W -o p --output-format csv -- python3 bench.py --workload $W --steps 5 --warmup 1 --no-cpu-baseline --pmc off --no-hbm-streaming > $O/${W}_under_rocprof.log 2>&1; echo "rocprof $W rc=$?"
cp $O/prof_$W/p_kernel_stats.csv $O/${W}_kernel_stats.csv 2>/dev/null; grep '^{' $O/${W}_under_rocprof.log | tail -1 > $O/${W}_bench_under_rocprof.json; head -5 $O/${W}_kernel_stats.csv | cut -c1-170; rm -rf $O/prof_$W
timeout -k 10 400 python3 bench.py > $O/bench_default_run.log 2>&1; echo "default rc=$?"; tail -c 600 $O/bench_default_run.log
