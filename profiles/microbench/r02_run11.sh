set -o pipefail
mkdir -p gpurun_out/r02j
O=gpurun_out/r02j
for MB in 32768 131072 262144 524288; do
  timeout -k 10 300 python bench.py --workload hqc128_mc --trials 1048576 --warmup 1 --mc-batch $MB > $O/mc_$MB.log 2> $O/mc_$MB.err || echo "mc $MB failed"
  python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/mc_$MB.log") if l.startswith("{")][-1])
    print("mc-batch $MB", "trials/s %.0f"%d["trials_per_s"], "value %.4g"%d["value"], "succ", d["decode_success_rate"], "chk", d["success_checksum"], "wall", round(d["wall_s"],3))
except Exception as e: print("mc-batch $MB ERR", e)
PY
done
