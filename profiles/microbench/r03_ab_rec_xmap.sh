#!/bin/bash
# round 3 A/B (second time, now that the pair is fabric-bound): record form, XCD-aware tile placement of the variable pass (SCALDPC_REC_XMAP)
O=gpurun_out/r03bc; mkdir -p $O
SCALDPC_REC_XMAP=1 timeout -k 10 300 python -m pytest tests/test_bp_gpu.py -q -m gpu -x -p no:cacheprovider -k "min_sum or minsum or record" > $O/pytest.log 2>&1; echo "pytest (xmap=1) rc=$?"; tail -2 $O/pytest.log
for V in 1 0 1 0; do
  P=off; [ $V$R = 1 ] && P=off
  SCALDPC_REC_XMAP=$V timeout -k 10 300 python bench.py --workload hqc128_minsum --steps 6 --warmup 2 --no-cpu-baseline --pmc live --no-hbm-streaming > $O/m$V.json 2> $O/b.err; echo "hqc128_minsum rec_xmap=$V rc=$?"
  python - <<PY
import json
d=json.loads([l for l in open("$O/m$V.json") if l.startswith("{")][-1])
t=d["roofline"].get("traffic_all_kernels") or {}
print("  ms/step %.3f  value %.4g  parity_ok %s  kernel_ms %s" % (d["ms_per_step"], d["value"], d["parity_ok"], d.get("kernel_ms")))
for k,v in t.items(): print("     ", k, "fetch %.1f MB write %.1f MB" % (v["fetch_bytes"]/1e6, v["write_bytes"]/1e6))
PY
done
