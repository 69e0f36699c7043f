set -o pipefail
mkdir -p gpurun_out/r02s
O=gpurun_out/r02s
B="--steps 5 --warmup 1 --no-cpu-baseline --pmc off --no-hbm-streaming"
for W in hqc256_tanh hqc128_tanh; do
  for V in "auto:" "occ3:SCALDPC_TANH_OCC=3"; do
    name=${V%%:*}; envs=${V#*:}
    env $envs timeout -k 10 200 python bench.py $B --workload $W > $O/ab_${W}_${name}.json 2> $O/ab_${W}_${name}.err || echo "bench $W $name failed"
    python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/ab_${W}_${name}.json") if l.startswith("{")][-1])
    r=d["roofline"]
    print("$W $name", "ms/step %.2f"%d["ms_per_step"], "frac %.4f"%r["frac"], {k:round(v["us"],2) for k,v in r["per_launch"].items()}, "iso", {k:round(v["us"],2) for k,v in r.get("isolated",{}).items()}, "parity", d.get("parity_ok"))
except Exception as e: print("$W $name", "ERR", e)
PY
  done
done
