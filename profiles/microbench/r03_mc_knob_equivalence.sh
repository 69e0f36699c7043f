#!/bin/bash
# round 3: the config-5 sweep at several noise levels, new schedule (defaults) vs everything new switched off:
# success_checksum / success rate / mean iterations must be identical; the line's own parity leg (oracle with early
# exit on the sweep's first trials) must pass.
O=gpurun_out/r03r; mkdir -p $O; : > $O/mc_knob_equivalence.log
for EPS in 0.0 0.02 0.05 0.08; do
  for V in on off; do
    if [ $V = on ]; then unset SCALDPC_FIRST_FUSED SCALDPC_FUSE_TEST; else export SCALDPC_FIRST_FUSED=0 SCALDPC_FUSE_TEST=0; fi
    timeout -k 10 300 python bench.py --workload hqc128_mc --trials 262144 --eps $EPS --warmup 1 --no-cpu-baseline --cpu-seconds 3 > $O/mc_${EPS}_$V.json 2> $O/err; rc=$?
    python - <<PY >> $O/mc_knob_equivalence.log
import json
d=json.loads([l for l in open("$O/mc_${EPS}_$V.json") if l.startswith("{")][-1])
print("eps $EPS new-schedule $V rc=$rc  trials/s %.0f  checksum %d  success %.6f  mean_iter %.6f  parity_ok %s (%d of %d iteration counts equal the oracle's)" % (d["trials_per_s"], d["success_checksum"], d["decode_success_rate"], d["mean_iterations"], d.get("parity_ok"), d.get("parity_same_iteration_count", -1), d.get("parity_checked", -1)))
PY
  done
done
cat $O/mc_knob_equivalence.log
