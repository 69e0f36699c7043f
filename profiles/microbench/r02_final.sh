set -o pipefail
mkdir -p gpurun_out/r02z
O=gpurun_out/r02z
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1; echo "all tests rc=$?" | tee -a $O/pytest_all.log; tail -4 $O/pytest_all.log
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout -k 10 400 python bench.py > $O/bench_default.log 2> $O/bench_default.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads([l for l in open("$O/bench_default.log") if l.startswith("{")][-1])
r=d["roofline"]
print({k:d[k] for k in ("value","ms_per_step","parity_ok","parity_checked","rccl_ranks")})
print("roofline", r["kernel"], r["bound"], round(r["achieved"]), round(r["frac"],4), "traffic", r["traffic"], r["traffic_source"], "hbm_stream", round(r["hbm_streaming_GBps"]), "dominant", r["dominant"])
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["reference_form_single_thread"]["value"])
PY
