#!/bin/bash
# round 3: lanes x tile-group sweep at the final state of the record form (CONFIGS="split group;split group;...")
O=gpurun_out/r03be; mkdir -p $O
IFS=";" read -ra CS <<< "${CONFIGS:-2 0;3 0;3 6;2 0;3 3}"; for C in "${CS[@]}"; do IFS=" " read -r A B <<< "$C"; set -- $A $B
  SCALDPC_SPLIT=$1 timeout -k 10 200 python bench.py --workload hqc128_minsum --steps 5 --warmup 1 --no-cpu-baseline --pmc off --no-hbm-streaming --tile-group $2 > $O/s$1_g$2.json 2> $O/b.err || { echo "split=$1 group=$2 failed"; continue; }
  python - <<PY
import json
d=json.loads([l for l in open("$O/s$1_g$2.json") if l.startswith("{")][-1])
print("split=$1 group=$2  ms/step %.3f  value %.4g  parity_ok %s kernel_ms %s" % (d["ms_per_step"], d["value"], d["parity_ok"], d.get("kernel_ms")))
PY
done
