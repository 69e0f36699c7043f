#!/bin/bash
# round 3 A/B: record-form variable pass -- XCD-aware tile placement (SCALDPC_REC_XMAP) x sc1 message stores (SCALDPC_REC_SC1)
O=gpurun_out/r03ai; mkdir -p $O
for X in 0 1; do for S in 0 1; do
  SCALDPC_REC_XMAP=$X SCALDPC_REC_SC1=$S timeout -k 10 300 python bench.py --workload hqc128_minsum --steps 8 --warmup 2 --no-cpu-baseline --pmc ${PMC:-off} --no-hbm-streaming > $O/x${X}_s$S.json 2> $O/b.err; echo "xmap=$X sc1=$S rc=$?"
  python - <<PY
import json
d=json.loads([l for l in open("$O/x${X}_s$S.json") if l.startswith("{")][-1])
t=d["roofline"].get("traffic_all_kernels") or {}
print("  ms/step %.3f  value %.4g  parity_ok %s  kernel_ms %s" % (d["ms_per_step"], d["value"], d["parity_ok"], d.get("kernel_ms")))
for k,v in t.items(): print("     ", k, "fetch %.1f MB write %.1f MB" % (v["fetch_bytes"]/1e6, v["write_bytes"]/1e6))
PY
done; done
