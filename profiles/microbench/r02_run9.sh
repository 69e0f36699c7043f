set -o pipefail
mkdir -p gpurun_out/r02h
O=gpurun_out/r02h
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/pytest_all.log 2>&1; echo "all tests rc=$?" | tee -a $O/pytest_all.log; tail -4 $O/pytest_all.log
timeout -k 10 200 python profiles/microbench/single_decode_converging.py > $O/single_decode.log 2>&1; tail -1 $O/single_decode.log
timeout -k 10 200 python profiles/microbench/single_decode_profile.py > $O/single_decode_nonconv.log 2>&1; tail -1 $O/single_decode_nonconv.log
timeout -k 10 300 python profiles/microbench/attack_loop_step.py > $O/attack_loop_step.log 2>&1; tail -2 $O/attack_loop_step.log
timeout -k 10 300 python profiles/microbench/small_batch_latency.py > $O/small_batch_latency.log 2>&1; tail -25 $O/small_batch_latency.log
