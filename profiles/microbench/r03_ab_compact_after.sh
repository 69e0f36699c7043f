#!/bin/bash
# round 3: with iteration 1 cheaper (first_fused), does the hand-over point of the compact pass move?
O=gpurun_out/r03k; mkdir -p $O
for V in 4 3 5 4 3 5; do
  SCALDPC_COMPACT_AFTER=$V timeout -k 10 300 python bench.py --workload hqc128_mc --trials 1000000 --warmup 1 --no-cpu-baseline --parity-rows 0 > $O/mc_ca$V.json 2> $O/mc.err; echo "compact_after=$V rc=$?"
  python - <<PY
import json
d=json.loads([l for l in open("$O/mc_ca$V.json") if l.startswith("{")][-1])
print("  trials/s %.0f  wall %.3f s  checksum %d" % (d["trials_per_s"], d["wall_s"], d["success_checksum"]))
PY
done
