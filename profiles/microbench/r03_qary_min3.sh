#!/bin/bash
# round 3: q-ary after the v_min3 folding in the unrolled check kernel: parity tests + config 4 + Kyber lines
O=gpurun_out/r03ay; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_qary_gpu.py tests/test_qary_property_gpu.py tests/test_exact_inference_gpu.py -q -m gpu -x -p no:cacheprovider > $O/pytest_qary.log 2>&1; echo "pytest rc=$?"; grep -v Deprec $O/pytest_qary.log | tail -2
for R in 1 2 3; do
  timeout -k 10 200 python bench.py --workload qary_config4 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]);k=d['kernel_ms'];print('config4 run $R  check %.1f us  var %.1f us  call %.3f ms  frac %.3f  parity_ok %s'%(k['check_per_launch']*1e3,k['var_per_launch']*1e3,d['ms_per_step'],d['roofline']['frac'],d['parity_ok']))"
done
