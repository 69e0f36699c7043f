#!/bin/bash
# round 3 A/B: launch order of the columns (SCALDPC_VAR_ORDER) under the min-sum record form, sc1 stores on/off
O=gpurun_out/r03aj; mkdir -p $O
for S in 0 1; do for V in ${ORDERS:-0 1 2 3}; do
  SCALDPC_VAR_ORDER=$V SCALDPC_REC_XMAP=0 SCALDPC_REC_SC1=$S timeout -k 10 300 python bench.py --workload hqc128_minsum --steps 8 --warmup 2 --no-cpu-baseline --pmc ${PMC:-off} --no-hbm-streaming > $O/v${V}_s$S.json 2> $O/b.err; echo "var_order=$V sc1=$S rc=$?"
  python - <<PY
import json
d=json.loads([l for l in open("$O/v${V}_s$S.json") if l.startswith("{")][-1])
t=d["roofline"].get("traffic_all_kernels") or {}
print("  ms/step %.3f  value %.4g  parity_ok %s  kernel_ms %s" % (d["ms_per_step"], d["value"], d["parity_ok"], d.get("kernel_ms")))
for k,v in t.items(): print("     ", k, "fetch %.1f MB write %.1f MB" % (v["fetch_bytes"]/1e6, v["write_bytes"]/1e6))
PY
done; done
