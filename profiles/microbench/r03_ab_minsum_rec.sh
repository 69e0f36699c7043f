#!/bin/bash
# round 3 A/B: min-sum in its record form (SCALDPC_MINSUM_REC=1) vs the message form (=0)
O=gpurun_out/r03af; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_bp_gpu.py -q -m gpu -x -p no:cacheprovider -k "min_sum or minsum or invisible or bench_configuration or config" > $O/pytest_minsum.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_minsum.log
for V in 1 0 1 0; do
  SCALDPC_MINSUM_REC=$V timeout -k 10 300 python bench.py --workload hqc128_minsum --steps 10 --warmup 2 --no-cpu-baseline --pmc off --no-hbm-streaming > $O/minsum_rec$V.json 2> $O/b.err; echo "hqc128_minsum rec=$V rc=$?"
  python - <<PY
import json
d=json.loads([l for l in open("$O/minsum_rec$V.json") if l.startswith("{")][-1])
print("  ms/step %.3f  value %.4g  parity_ok %s  success %.6f  kernel_ms %s" % (d["ms_per_step"], d["value"], d["parity_ok"], d["decode_success_rate"], d.get("kernel_ms")))
PY
done
