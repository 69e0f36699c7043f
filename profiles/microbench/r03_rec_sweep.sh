#!/bin/bash
# round 3: lanes x tile-group sweep of the min-sum record form on the HQC-128 bench
O=gpurun_out/r03ag; mkdir -p $O
for S in 1 2 3; do for G in 0 2 4 6 8 12; do
  SCALDPC_SPLIT=$S timeout -k 10 200 python bench.py --workload hqc128_minsum --steps 5 --warmup 1 --no-cpu-baseline --pmc off --no-hbm-streaming --tile-group $G > $O/s${S}_g$G.json 2> $O/b.err || { echo "split=$S group=$G failed"; tail -3 $O/b.err; continue; }
  python - <<PY
import json
d=json.loads([l for l in open("$O/s${S}_g$G.json") if l.startswith("{")][-1])
print("split=$S group=$G  ms/step %.3f  value %.4g  parity_ok %s kernel_ms %s" % (d["ms_per_step"], d["value"], d["parity_ok"], d.get("kernel_ms")))
PY
done; done
