O=gpurun_out/r03an; mkdir -p $O
for X in 3 4; do
  cp profiles/microbench/libscaldpc_x$X.so sca-ldpc_amd/libscaldpc.so
  timeout -k 10 300 python bench.py --workload hqc128_minsum --steps 5 --warmup 1 --no-cpu-baseline --pmc off --no-hbm-streaming --parity-rows 0 > $O/x$X.json 2> $O/b.err; echo "experiment=$X rc=$?"
  python - <<PY
import json
d=json.loads([l for l in open("$O/x$X.json") if l.startswith("{")][-1])
print("  ms/step %.3f  kernel_ms %s  isolated %s" % (d["ms_per_step"], d.get("kernel_ms"), {k:round(v["us"],1) for k,v in d["roofline"].get("isolated",{}).items()}))
PY
done
