#!/bin/bash
# round 3: where do the variable pass's record fetches come from?  Live PMC bytes per launch of k_var_rec with one / two
# stream lanes and 1 / 2 / 4 tiles per group (fewer tiles in flight = fewer record planes per L2; one lane = no check pass
# streaming beside it)
O=gpurun_out/r03av; mkdir -p $O
for C in "2 0" "1 4" "1 2" "1 1" "2 2"; do set -- $C
  SCALDPC_SPLIT=$1 timeout -k 10 300 python bench.py --workload hqc128_minsum --steps 4 --warmup 1 --no-cpu-baseline --pmc live --no-hbm-streaming --tile-group $2 > $O/s$1_g$2.json 2> $O/b.err; echo "split=$1 group=$2 rc=$?"
  python - <<PY
import json
d=json.loads([l for l in open("$O/s$1_g$2.json") if l.startswith("{")][-1])
t=d["roofline"].get("traffic_all_kernels") or {}
print("  ms/step %.3f  kernel_ms %s" % (d["ms_per_step"], d.get("kernel_ms")))
for k,v in t.items(): print("     ", k, "fetch %.1f MB write %.1f MB dispatches %d" % (v["fetch_bytes"]/1e6, v["write_bytes"]/1e6, v["dispatches"]))
PY
done
