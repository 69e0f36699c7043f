set -o pipefail
mkdir -p gpurun_out/r02c
O=gpurun_out/r02c
timeout -k 10 600 python -m pytest tests/test_append_gpu.py tests/test_call_protocol_gpu.py -m gpu -x -q > $O/pytest_new.log 2>&1; echo "new tests rc=$?" | tee -a $O/pytest_new.log
tail -5 $O/pytest_new.log
grep -q "rc=0" $O/pytest_new.log || echo "NEW TESTS FAILED (continuing)"
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1; echo "all tests rc=$?" | tee -a $O/pytest_all.log
tail -8 $O/pytest_all.log
B="--steps 5 --warmup 1 --no-cpu-baseline --pmc off --no-hbm-streaming"
for W in hqc128_minsum hqc256_tanh; do
  for V in "base:" "form1:SCALDPC_VAR_FORM=1" "order1:SCALDPC_VAR_ORDER=1" "both:SCALDPC_VAR_FORM=1 SCALDPC_VAR_ORDER=1"; do
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
