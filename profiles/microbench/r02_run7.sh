set -o pipefail
mkdir -p gpurun_out/r02f
O=gpurun_out/r02f
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1; echo "all tests rc=$?" | tee -a $O/pytest_all.log
tail -4 $O/pytest_all.log
timeout -k 10 300 python profiles/microbench/attack_loop_step.py > $O/attack_loop_step.log 2>&1; tail -2 $O/attack_loop_step.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_kyber -o kyber --output-format csv -- python3 profiles/microbench/kyber_check.py 5 tree > $O/prof_kyber.log 2>&1; echo "prof kyber rc=$?"; tail -4 $O/prof_kyber.log
: > $O/all_workloads.log
for W in hqc128_minsum hqc128_tanh hqc192_minsum hqc256_tanh; do
  timeout -k 10 400 python bench.py --workload $W --steps 5 --warmup 1 --no-cpu-baseline >> $O/all_workloads.log 2> $O/bench_$W.err || echo "bench $W failed"
done
timeout -k 10 300 python bench.py --workload hqc128_mc --trials 1000000 --warmup 1 >> $O/all_workloads.log 2> $O/bench_mc.err || echo "bench mc failed"
timeout -k 10 300 python bench.py --workload qary_config4 --steps 20 --warmup 2 >> $O/all_workloads.log 2> $O/bench_q4.err || echo "bench q4 failed"
python - <<PY
import json
for l in open("$O/all_workloads.log"):
    if not l.startswith("{"): continue
    d=json.loads(l); r=d.get("roofline",{})
    print(d["config"]["workload"][:60], "| value %.4g"%d["value"], "ms/step %.2f"%d["ms_per_step"], "frac", round(r.get("frac",0),4), "traffic", r.get("traffic"), "alg", r.get("algorithmic_bytes_per_launch"), "hbm_stream", r.get("hbm_streaming_GBps"), "parity", d.get("parity_ok"))
PY
for W in hqc128_minsum hqc256_tanh; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM --output-format csv -d $O/sq_$W -o sq -- python3 bench.py --pmc-child --workload $W --batch 256 --pmc off > $O/sq_$W.log 2>&1; echo "sq $W rc=$?"
  f=$(find $O/sq_$W -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python profiles/sq_summarise.py $f $O/sq_counters_$W.json
done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*counter_collection.csv" -delete
