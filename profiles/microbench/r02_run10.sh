set -o pipefail
mkdir -p gpurun_out/r02i
O=gpurun_out/r02i
export TMPDIR=/tmp
SCALDPC_POISON=1 timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/pytest_poison.log 2>&1; echo "poison suite rc=$?" | tee -a $O/pytest_poison.log; tail -4 $O/pytest_poison.log
SCALDPC_PROPERTY_EXAMPLES=400 timeout -k 10 900 python -m pytest tests/test_append_gpu.py -m gpu -q -k random_append > $O/pytest_append_property.log 2>&1; echo "append property rc=$?" | tee -a $O/pytest_append_property.log; tail -4 $O/pytest_append_property.log
timeout -k 10 300 python profiles/microbench/small_batch_latency.py > $O/small_batch_latency.log 2>&1; tail -22 $O/small_batch_latency.log
