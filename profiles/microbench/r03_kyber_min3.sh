#!/bin/bash
# round 3: Kyber DecoderSpecial tree kernel with the shared minima taking two candidates per v_min3_f32: parity tests + bench lines
O=gpurun_out/r03bf; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_qary_gpu.py tests/test_qary_property_gpu.py tests/test_exact_inference_gpu.py -q -m gpu -x -p no:cacheprovider > $O/pytest_qary.log 2>&1; echo "pytest rc=$?"; grep -v Deprec $O/pytest_qary.log | tail -2
for B in 256 64 1; do for R in 1 2; do
  timeout -k 10 200 python bench.py --workload kyber_sw6 --batch $B --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]);print('kyber_sw6 batch $B run $R  call %.3f ms  value %.4g  frac %s  parity_ok %s'%(d['ms_per_step'],d['value'],d['roofline'].get('frac'),d['parity_ok']))"
done; done
