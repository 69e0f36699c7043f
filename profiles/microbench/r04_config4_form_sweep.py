"""Device-loop time of one DecoderN450R150V3C7B1 call (config 4's decoder, 5 iterations) by batch size: unrolled enumeration
(k_q_check_unrolled<3,7>) against the clipped min-plus recursion (k_q_check_dp<3,7>); HIP events on the handle's stream.
    python profiles/microbench/r04_config4_form_sweep.py > gpurun_out/.../config4_form_sweep.log"""
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
S = importlib.import_module("sca-ldpc_amd")
qary = importlib.import_module("sca-ldpc_amd.qary")
gens = json.load(open(os.path.join(ROOT, "tests", "golden", "generators.json")))
g = S.TannerGraph.from_coo(gens["regular_identity_300_150_3_6_s1"])
dec = qary.decoder_class("DecoderN450R150V3C7B1")(g.to_dense(np.int8), 5)
rng = np.random.RandomState(3)
p = 1 / 3
good, bad = np.array([p, 1.75 * p, 0.25 * p]), np.array([p, 0.25 * p, 1.75 * p])
for batch in (1, 64, 256, 1024, 4096, 16384):
    pmf = np.where((rng.rand(batch, g.n) < 0.02)[:, :, None], bad, good).astype(np.float32)
    row, outs = {}, {}
    for form, kn in (("unrolled", dict(dp=0)), ("dp", dict(dp=1))):
        dec.configure(timing=1, **kn)
        best = None
        for _ in range(6):
            outs[form] = dec.min_sum_batch(pmf)
            t = dec.last_timing()
            if best is None or t["ms_loop"] < best["ms_loop"]:
                best = t
        row[form] = best
    assert np.array_equal(outs["unrolled"], outs["dp"])
    print(f"batch {batch:6d}  {row['unrolled']['check_kernel']}: loop {row['unrolled']['ms_loop']:.3f} ms (check {row['unrolled']['ms_check']:.3f})   "
          f"{row['dp']['check_kernel']}: loop {row['dp']['ms_loop']:.3f} ms (check {row['dp']['ms_check']:.3f})   same symbols", flush=True)
dec.close()
