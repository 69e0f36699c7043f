set -o pipefail
mkdir -p gpurun_out/r02g
O=gpurun_out/r02g
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_qary_gpu.py tests/test_append_gpu.py -m gpu -q > $O/pytest_sel.log 2>&1; echo "sel tests rc=$?" | tee -a $O/pytest_sel.log; tail -3 $O/pytest_sel.log
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout -k 10 200 python profiles/microbench/single_decode_converging.py > $O/single_decode.log 2>&1; tail -1 $O/single_decode.log
timeout -k 10 300 rocprofv3 --hip-trace --kernel-trace --stats -d $O/prof_single -o single --output-format csv -- python3 profiles/microbench/single_decode_converging.py > $O/prof_single.log 2>&1; echo "prof single rc=$?"
for f in $(find $O/prof_single -name "*hip_api_stats.csv" -o -name "*kernel_stats.csv"); do echo $f; head -14 $f | cut -c1-160; done
timeout -k 10 300 python bench.py > $O/bench_default.log 2> $O/bench_default.err; echo "bench rc=$?"; python - <<PY
import json
d=json.loads([l for l in open("$O/bench_default.log") if l.startswith("{")][-1])
print({k:d[k] for k in ("value","ms_per_step","parity_ok","rccl_ranks")}, d["roofline"]["frac"], d["roofline"]["traffic"], d["cpu_baseline"]["value"], d["cpu_baseline"]["reference_form_single_thread"])
PY
find $O -name "*_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
