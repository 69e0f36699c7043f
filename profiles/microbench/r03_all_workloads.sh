#!/bin/bash
# round 3, end of round: every HQC workload's bench line with the final library -> gpurun_out/r03p/all_workloads.log
O=gpurun_out/r03p; mkdir -p $O; : > $O/all_workloads.log
for W in hqc128_minsum hqc128_tanh hqc192_minsum hqc256_tanh; do
  timeout -k 10 400 python bench.py --workload $W --steps 5 --warmup 1 --no-cpu-baseline > $O/$W.json 2> $O/$W.err; echo "$W rc=$?"
  grep '^{' $O/$W.json | tail -1 >> $O/all_workloads.log
done
python - <<'PY'
import json
for l in open("gpurun_out/r03p/all_workloads.log"):
    d=json.loads(l); r=d["roofline"]
    print(d["config"]["workload"][:60].ljust(62), "%.4g upd/s"%d["value"], "%.2f ms"%d["ms_per_step"], "whole-job frac %.3f"%(d["whole_job_algorithmic_GBps"]/8000), "pair %.3f"%r["frac"], "dominant alone", r["dominant"]["frac"] and round(r["dominant"]["frac"],3), "traffic x%.3f"%(r["traffic"]/r["algorithmic_bytes_per_launch"]) if r["traffic"] else "", d["parity_ok"])
PY
