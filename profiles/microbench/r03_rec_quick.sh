#!/bin/bash
# round 3: quick look at the min-sum record form on the HQC-128 bench (3 runs, live PMC on the last)
O=gpurun_out/r03am; mkdir -p $O
for R in 1 2 3; do
  P=off; [ $R = 3 ] && P=live
  timeout -k 10 300 python bench.py --workload hqc128_minsum --steps 8 --warmup 2 --no-cpu-baseline --pmc $P --no-hbm-streaming > $O/run$R.json 2> $O/b.err; echo "run $R rc=$?"
  python - <<PY
import json
d=json.loads([l for l in open("$O/run$R.json") if l.startswith("{")][-1])
t=d["roofline"].get("traffic_all_kernels") or {}
print("  ms/step %.3f  value %.4g  parity_ok %s  kernel_ms %s" % (d["ms_per_step"], d["value"], d["parity_ok"], d.get("kernel_ms")))
for k,v in t.items(): print("     ", k, "fetch %.1f MB write %.1f MB" % (v["fetch_bytes"]/1e6, v["write_bytes"]/1e6))
PY
done
