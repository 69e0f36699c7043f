#!/usr/bin/env python3
"""DecoderN1280R512SW6 (the Kyber decoder "used in the paper", simulate/kyber.py:381-402; lib.rs:66-75) on
one MI355X: time per min_sum call at batch 1 / 16 / 64 / 256 for the three check-kernel families
(tree walk = default, generic wave-parallel enumeration, codeword-per-lane enumeration), identical
outputs, and the tree kernel's place on the VALU roofline.

Work of one check update (decoder_special.rs:506-563): 5^6 = 15 625 assignments x (7 adds for S + 7
subtractions + 7 minima) = 328 125 f32 operations as the reference performs them; the tree kernel shares
prefix sums and executes 2 + 7 + 7 = 16 per assignment = 250 000.  512 checks per iteration.
VALU peak for add / min (one operation per lane per 2 cycles per SIMD... 157.3 TFLOP/s counts an FMA as
two): 78.6e12 operations/s."""
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
S = importlib.import_module("sca-ldpc_amd")
qary = importlib.import_module("sca-ldpc_amd.qary")
PEAK_OPS = 78.6e12


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    only_tree = len(sys.argv) > 2 and sys.argv[2] == "tree"  # (for a kernel profile of the default path alone)
    gens = json.load(open(os.path.join(ROOT, "tests", "golden", "generators.json")))
    g = S.TannerGraph.from_coo(gens["qary_qc_256_6_3_s0_cb2"])
    H = g.to_dense(np.int8)
    rng = np.random.RandomState(10)
    dec = qary.decoder_class("DecoderN1280R512SW6")(H, iters)
    for batch in ((256,) if len(sys.argv) > 3 else (1, 16, 64, 256)):  # (a third argument: batch 256 only, for counter passes)
        pb = rng.dirichlet(np.ones(5), size=(batch, 768)).astype(np.float32)
        ps = rng.dirichlet(np.ones(25), size=(batch, 512)).astype(np.float32)
        res, out = {}, {}
        for name, kn in (("tree", dict(wave=-1, tree=1)), ("generic_wave", dict(wave=1, tree=0)), ("lane", dict(wave=0, tree=0))):
            if (name == "lane" and batch < 64) or (only_tree and name != "tree"):
                continue  # (one codeword per lane: pointless below a wave's worth)
            dec.configure(**kn)
            dec.min_sum_batch(pb, ps)
            reps = 3 if name != "tree" else 10
            t0 = time.perf_counter()
            for _ in range(reps):
                out[name] = dec.min_sum_batch(pb, ps)
            res[name] = (time.perf_counter() - t0) / reps
        same = all(np.array_equal(out["tree"], o) for o in out.values())
        ops_exec = 512.0 * batch * iters * 15625 * 16
        print(json.dumps({"batch": batch, "iterations": iters, "ms_per_call": {k: v * 1e3 for k, v in res.items()},
                          "outputs_identical": same,
                          "tree_executed_valu_ops_per_s": ops_exec / res["tree"], "tree_frac_of_valu_peak": ops_exec / res["tree"] / PEAK_OPS,
                          "reference_ops_per_s": 512.0 * batch * iters * 15625 * 21 / res["tree"]}))
    dec.close()


if __name__ == "__main__":
    main()
