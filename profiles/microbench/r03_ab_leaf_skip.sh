#!/bin/bash
# round 3 A/B: tanh rule (message form), fixed iterations: passes without output leave out the columns of degree <= 1 and the
# (REJECTED and reverted: the knob SCALDPC_LEAF_SKIP does not exist in the library; see profiles/r03/ab_leaf_skip_rejected.log)
# check pass does not store the messages into them (SCALDPC_LEAF_SKIP)
O=gpurun_out/r03bi; mkdir -p $O
SCALDPC_LEAF_SKIP=1 timeout -k 10 400 python -m pytest tests/test_bp_gpu.py tests/test_exact_inference_gpu.py -q -m gpu -x -p no:cacheprovider -k "not min_sum" > $O/pytest.log 2>&1; echo "pytest (leaf_skip=1) rc=$?"; grep -v Deprec $O/pytest.log | tail -2
for W in hqc256_tanh hqc128_tanh; do for V in 1 0 1 0; do
  SCALDPC_LEAF_SKIP=$V timeout -k 10 300 python bench.py --workload $W --steps 5 --warmup 1 --no-cpu-baseline --pmc off --no-hbm-streaming > $O/${W}_$V.json 2> $O/b.err; echo "$W leaf_skip=$V rc=$?"
  python - <<PY
import json
d=json.loads([l for l in open("$O/${W}_$V.json") if l.startswith("{")][-1])
print("  ms/step %.3f  value %.4g  parity_ok %s  kernel_ms %s" % (d["ms_per_step"], d["value"], d["parity_ok"], d.get("kernel_ms")))
PY
done; done
