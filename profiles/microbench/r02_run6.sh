set -o pipefail
mkdir -p gpurun_out/r02e
O=gpurun_out/r02e
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 200 python profiles/microbench/attack_loop_step.py > $O/attack_loop_step.log 2>&1; tail -3 $O/attack_loop_step.log
SCALDPC_TIMING=1 timeout -k 10 200 python profiles/microbench/attack_loop_step.py 2>&1 | grep "app:" | tail -12 > $O/append_phases.log; cat $O/append_phases.log
timeout -k 10 200 python bench.py --workload qary_config4 --steps 20 --warmup 2 > $O/bench_qary_config4.log 2>&1; tail -1 $O/bench_qary_config4.log
BENCH_FORCE_DEVICE=0 BENCH_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_gpus2_gloo.log 2>&1; echo "gpus2 rc=$?"; grep '^{' $O/bench_gpus2_gloo.log | tail -1 | cut -c1-600
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_kyber -o kyber --output-format csv -- python3 profiles/microbench/kyber_check.py 5 > $O/prof_kyber.log 2>&1; echo "prof kyber rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_q4 -o q4 --output-format csv -- python3 bench.py --workload qary_config4 --steps 20 --warmup 2 > $O/prof_q4.log 2>&1; echo "prof q4 rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_default -o def --output-format csv -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --pmc off --no-hbm-streaming > $O/prof_default.log 2>&1; echo "prof default rc=$?"
find $O -name "*kernel_stats.csv" | head; for f in $(find $O -name "*kernel_stats.csv"); do echo $f; head -8 $f | cut -c1-200; done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
