#!/bin/bash
# round 3 A/B: record form, passes without output leave out the columns of degree <= 1 (SCALDPC_REC_SKIP1)
O=gpurun_out/r03az; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_bp_gpu.py -q -m gpu -x -p no:cacheprovider -k "min_sum or minsum or record" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
for W in hqc128_minsum hqc192_minsum; do for V in 1 0 1 0; do
  SCALDPC_REC_SKIP1=$V timeout -k 10 300 python bench.py --workload $W --steps 6 --warmup 2 --no-cpu-baseline --pmc off --no-hbm-streaming > $O/${W}_$V.json 2> $O/b.err; echo "$W rec_skip1=$V rc=$?"
  python - <<PY
import json
d=json.loads([l for l in open("$O/${W}_$V.json") if l.startswith("{")][-1])
print("  ms/step %.3f  value %.4g  parity_ok %s  kernel_ms %s" % (d["ms_per_step"], d["value"], d["parity_ok"], d.get("kernel_ms")))
PY
done; done
