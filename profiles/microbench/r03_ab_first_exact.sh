#!/bin/bash
# round 3 A/B: k_var_first as exact-degree straight-line code (SCALDPC_FIRST_EXACT=1) vs the bucketed form (=0)
# (the knob SCALDPC_FIRST_EXACT existed only for this measurement: no gain, the variant was removed again)
O=gpurun_out/r03ax; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_bp_gpu.py -q -m gpu -x -p no:cacheprovider -k "first_iteration or riding or record_form or random_graph or infinite" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
for V in 1 0 1 0; do
  SCALDPC_FIRST_EXACT=$V timeout -k 10 300 python bench.py --workload hqc128_mc --trials 1000000 --warmup 1 --no-cpu-baseline --parity-rows 0 > $O/mc_fx$V.json 2> $O/mc.err; echo "mc first_exact=$V rc=$?"
  python - <<PY
import json
d=json.loads([l for l in open("$O/mc_fx$V.json") if l.startswith("{")][-1])
print("  trials/s %.0f  wall %.3f s  checksum %d  success %.6f  mean_iter %.6f" % (d["trials_per_s"], d["wall_s"], d["success_checksum"], d["decode_success_rate"], d["mean_iterations"]))
PY
done
for W in hqc128_minsum hqc256_tanh; do for V in 1 0; do
  SCALDPC_FIRST_EXACT=$V timeout -k 10 300 python bench.py --workload $W --steps 6 --warmup 2 --no-cpu-baseline --pmc off --no-hbm-streaming > $O/${W}_fx$V.json 2> $O/b.err; echo "$W first_exact=$V rc=$?"
  python - <<PY
import json
d=json.loads([l for l in open("$O/${W}_fx$V.json") if l.startswith("{")][-1])
print("  ms/step %.3f  value %.4g  parity_ok %s" % (d["ms_per_step"], d["value"], d["parity_ok"]))
PY
done; done
