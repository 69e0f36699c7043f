set -o pipefail
mkdir -p gpurun_out/r02d
O=gpurun_out/r02d
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1; echo "all tests rc=$?" | tee -a $O/pytest_all.log
tail -6 $O/pytest_all.log
timeout -k 10 300 python profiles/microbench/kyber_check.py 5 > $O/kyber_check.log 2>&1; cat $O/kyber_check.log | tail -6
timeout -k 10 200 python profiles/microbench/attack_loop_step.py > $O/attack_loop_step.log 2>&1; tail -3 $O/attack_loop_step.log
B="--steps 5 --warmup 1 --no-cpu-baseline --pmc off --no-hbm-streaming"
for W in hqc256_tanh hqc128_tanh; do
  for V in "base:" "pipe256:SCALDPC_CHECK_PIPE=256" "pipe512:SCALDPC_CHECK_PIPE=512"; do
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
