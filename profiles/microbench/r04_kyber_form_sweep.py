"""Device-loop time of one DecoderN1280R512SW6 call (5 iterations) by batch size, check kernel form = tree walk against
the two min-plus recursions (HIP events on the handle's stream, scaldpc_qary_last_timing): where the library switches.
    python profiles/microbench/r04_kyber_form_sweep.py > gpurun_out/.../kyber_form_sweep.log"""
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
g = S.TannerGraph.from_coo(gens["qary_qc_256_6_3_s0_cb2"])
dec = qary.decoder_class("DecoderN1280R512SW6")(g.to_dense(np.int8), 5)
rng = np.random.RandomState(3)
for batch in (1, 2, 4, 8, 12, 16, 24, 32, 48, 64, 96, 128, 192, 256, 512, 1024):
    pb = rng.dirichlet(np.ones(5), size=(batch, 768)).astype(np.float32)
    ps = rng.dirichlet(np.ones(25), size=(batch, 512)).astype(np.float32)
    row, outs = {}, {}
    for form, kn in (("tree", dict(dp=0)), ("dp", dict(dp=1, dp_min=1, dp_split=0, dp_split2=0)), ("half", dict(dp=1, dp_min=1, dp_split=0, dp_split2=1 << 20)),
                     ("split", dict(dp=1, dp_min=1, dp_split=1 << 20))):
        dec.configure(timing=1, **kn)
        best = None
        for _ in range(6):
            outs[form] = dec.min_sum_batch(pb, ps)
            t = dec.last_timing()
            if best is None or t["ms_loop"] < best["ms_loop"]:
                best = t
        row[form] = best
    assert np.array_equal(outs["tree"], outs["dp"]) and np.array_equal(outs["tree"], outs["split"]) and np.array_equal(outs["tree"], outs["half"])
    print(f"batch {batch:5d}  tree: loop {row['tree']['ms_loop']:.3f} ms (check {row['tree']['ms_check']:.3f})   "
          f"min-plus, row per lane: loop {row['dp']['ms_loop']:.3f} ms (check {row['dp']['ms_check']:.3f})   "
          f"min-plus, row over two waves: loop {row['half']['ms_loop']:.3f} ms (check {row['half']['ms_check']:.3f})   "
          f"min-plus, row over four waves: loop {row['split']['ms_loop']:.3f} ms (check {row['split']['ms_check']:.3f})   same symbols", flush=True)
dec.close()
