#!/bin/bash
# round 3 A/B: the convergence test riding on the next check pass (SCALDPC_FUSE_TEST=1) vs the stand-alone launch (=0)
O=gpurun_out/r03m; mkdir -p $O
for V in 1 0 1 0; do
  SCALDPC_FUSE_TEST=$V timeout -k 10 300 python bench.py --workload hqc128_mc --trials 1000000 --warmup 1 --no-cpu-baseline --parity-rows 0 > $O/mc_ft$V.json 2> $O/mc.err; echo "mc fuse_test=$V rc=$?"
  python - <<PY
import json
d=json.loads([l for l in open("$O/mc_ft$V.json") if l.startswith("{")][-1])
print("  trials/s %.0f  wall %.3f s  checksum %d  success %.6f  mean_iter %.6f" % (d["trials_per_s"], d["wall_s"], d["success_checksum"], d["decode_success_rate"], d["mean_iterations"]))
PY
done
