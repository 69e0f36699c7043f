#!/bin/bash
# round 3 A/B: the unrolled q-ary check kernel compiled for 1 / 2 / 3 / 4 waves per SIMD (SCALDPC_QARY_OCC)
for B in 1024 2048 704; do for V in 1 2 3 4; do
  SCALDPC_QARY_OCC=$V timeout -k 10 200 python bench.py --workload qary_config4 --batch $B --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]);k=d['kernel_ms'];print('batch $B occ $V  check %.1f us  var %.1f us  call %.3f ms  frac %.3f  parity_ok %s'%(k['check_per_launch']*1e3,k['var_per_launch']*1e3,d['ms_per_step'],d['roofline']['frac'],d['parity_ok']))"
done; done
