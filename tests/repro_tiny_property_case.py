"""Test tooling (not collected by pytest): the tiny dense instance on which a 15 000-example soak of tests/test_bp_property_gpu.py found the
property test's tolerance too tight in round 4 (a 3000x chaotic amplification of one-ulp perturbations in the f32 tanh rule) -- HIP path
against the oracle, with the perturbation experiment that sized the new tolerance.  Lives under tests/ because it calls the oracle."""
import importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
S = importlib.import_module("sca-ldpc_amd"); bp = importlib.import_module("sca-ldpc_amd.bp")
from oracle import pyoracle as oracle
m, n, density, batch, max_iter, seed = 22, 23, 0.3267453472929817, 74, 10, 2
rng = np.random.RandomState(seed)
H = (rng.rand(m, n) < density).astype(np.int8)
g = S.TannerGraph.from_dense(H)
probs = rng.uniform(0.005, 0.3, size=n)
err = (rng.rand(batch, n) < np.maximum(probs, 0.02)[None, :]).astype(np.uint8)
x = g.syndrome(err)
for path in ("auto", "stream", "edge"):
    if path == "auto": os.environ.pop("SCALDPC_PATH", None)
    else: os.environ["SCALDPC_PATH"] = path
    dec = bp.bp_decoder(g, max_iter=max_iter, bp_method="product_sum", channel_probs=probs, input_vector_type="syndrome")
    got = dec.decode_batch(x, early_exit=False, want_llr=True, input_vector_type="syndrome"); dec.close()
    r32 = oracle.bp_decode_batch(g, probs, x, 0, max_iter, "tanh_complement", dtype="f32", threads=4, early_exit=False)
    r64 = oracle.bp_decode_batch(g, probs, x, 0, max_iter, "tanh_complement", dtype="f64", threads=4, early_exit=False)
    a, b, c = got["llr"].astype(np.float64), r32["llr"].astype(np.float64), r64["llr"]
    with np.errstate(invalid="ignore"):
        dev = np.abs(a - b) / (1 + np.abs(b)); own = np.abs(c - b) / (1 + np.abs(b))
    fin = np.isfinite(dev) & np.isfinite(own)
    i = np.unravel_index(np.nanargmax(np.where(fin, dev, -1)), dev.shape)
    print(path, "max device-vs-f32 rel", dev[fin].max(), "at", i, "dev", a[i], "o32", b[i], "o64", c[i], "| oracle own max", own[fin].max(), "tol", max(2e-4, 20 * own[fin].max()))
    cw = i[0]
    print("   codeword", cw, "own max on that codeword", own[cw][np.isfinite(own[cw])].max(), "device max on it", dev[cw][np.isfinite(dev[cw])].max())
