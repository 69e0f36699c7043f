#!/bin/bash
# round 3 A/B: iteration 1 without its check pass (SCALDPC_FIRST_FUSED=1, default) vs with it (=0)
O=gpurun_out/r03i; mkdir -p $O
for V in 1 0 1 0; do
  SCALDPC_FIRST_FUSED=$V timeout -k 10 300 python bench.py --workload hqc128_mc --trials 1000000 --warmup 1 --no-cpu-baseline --parity-rows 0 > $O/mc_ff$V.json 2> $O/mc.err; echo "mc first_fused=$V rc=$?"
  python - <<PY
import json
d=json.loads([l for l in open("$O/mc_ff$V.json") if l.startswith("{")][-1])
print("  trials/s %.0f  wall %.3f s  checksum %d  success %.6f  mean_iter %.6f" % (d["trials_per_s"], d["wall_s"], d["success_checksum"], d["decode_success_rate"], d["mean_iterations"]))
PY
done
for W in hqc128_minsum hqc256_tanh; do for V in 1 0; do
  SCALDPC_FIRST_FUSED=$V timeout -k 10 300 python bench.py --workload $W --steps 10 --warmup 2 --no-cpu-baseline --pmc off --no-hbm-streaming > $O/${W}_ff$V.json 2> $O/b.err; echo "$W first_fused=$V rc=$?"
  python - <<PY
import json
d=json.loads([l for l in open("$O/${W}_ff$V.json") if l.startswith("{")][-1])
print("  ms/step %.3f  value %.4g  parity_ok %s  success %.6f" % (d["ms_per_step"], d["value"], d["parity_ok"], d["decode_success_rate"]))
PY
done; done
