#!/bin/bash
# usage: run_ab.sh OUTDIR NAME...   (NAME = base or a variant under build/ab); one bench line per variant
out=$1; shift
mkdir -p $out
for v in "$@"; do
  if [ "$v" = base ]; then unset SCALDPC_SO; else export SCALDPC_SO=$PWD/build/ab/libscaldpc_$v.so; fi
  timeout -k 10 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-hbm-streaming > $out/$v.json 2> $out/$v.err || echo "FAILED $v"
  python3 - "$out/$v.json" "$v" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d["roofline"]; t=r.get("traffic_all_kernels",{})
print(sys.argv[2], "ms/step %.2f"%d["ms_per_step"], "pair us %.1f"%r["pair"]["us"], {k:round(v["us"],1) for k,v in r["per_launch"].items()},
      "bytes MB", {k:(round(v["fetch_bytes"]/1e6,1),round(v["write_bytes"]/1e6,1)) for k,v in t.items()}, "frac %.3f"%r["frac"], "parity", d.get("parity_ok"))
PY
done
