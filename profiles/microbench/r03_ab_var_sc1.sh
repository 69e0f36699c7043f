#!/bin/bash
# round 3 A/B: sc1 message stores in the variable pass -- record form (SCALDPC_REC_SC1, min-sum) and message form (SCALDPC_VAR_SC1, tanh rule)
O=gpurun_out/r03al; mkdir -p $O
for V in 1 0 1 0; do
  SCALDPC_REC_SC1=$V timeout -k 10 300 python bench.py --workload hqc128_minsum --steps 8 --warmup 2 --no-cpu-baseline --pmc off --no-hbm-streaming > $O/minsum_recsc1_$V.json 2> $O/b.err; echo "hqc128_minsum rec_sc1=$V rc=$?"
  python - <<PY
import json
d=json.loads([l for l in open("$O/minsum_recsc1_$V.json") if l.startswith("{")][-1])
print("  ms/step %.3f  value %.4g  parity_ok %s  kernel_ms %s" % (d["ms_per_step"], d["value"], d["parity_ok"], d.get("kernel_ms")))
PY
done
for W in hqc256_tanh hqc128_tanh; do for V in 1 0 1 0; do
  SCALDPC_VAR_SC1=$V timeout -k 10 300 python bench.py --workload $W --steps 6 --warmup 2 --no-cpu-baseline --pmc off --no-hbm-streaming > $O/${W}_sc1_$V.json 2> $O/b.err; echo "$W var_sc1=$V rc=$?"
  python - <<PY
import json
d=json.loads([l for l in open("$O/${W}_sc1_$V.json") if l.startswith("{")][-1])
print("  ms/step %.3f  value %.4g  parity_ok %s  kernel_ms %s" % (d["ms_per_step"], d["value"], d["parity_ok"], d.get("kernel_ms")))
PY
done; done
for V in 1 0 1 0; do
  SCALDPC_VAR_SC1=$V timeout -k 10 300 python bench.py --workload hqc128_mc --trials 1000000 --warmup 1 --no-cpu-baseline --parity-rows 0 > $O/mc_sc1_$V.json 2> $O/mc.err; echo "mc var_sc1=$V rc=$?"
  python - <<PY
import json
d=json.loads([l for l in open("$O/mc_sc1_$V.json") if l.startswith("{")][-1])
print("  trials/s %.0f  wall %.3f s  checksum %d  success %.6f  mean_iter %.6f" % (d["trials_per_s"], d["wall_s"], d["success_checksum"], d["decode_success_rate"], d["mean_iterations"]))
PY
done
