"""TEST INFRASTRUCTURE -- NOT PART OF THE PRODUCT PATH.

ctypes front end of oracle/liboracle.so (the CPU restatement of the reference's
decoders; see bp_oracle.c / qary_oracle.c for what each function follows and
its parity status).  May be imported only by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by anything under sca-ldpc_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

QERR = {
    -3: "pmf does not sum to 1 +- 1e-3 (decoder.rs:683-684)",
    -4: "No maximum probability found (decoder.rs:680)",
    -5: "a message has no finite entry (decoder.rs:368-375 would not terminate)",
    -6: "no valid configuration for a check (decoder.rs:618)",
    -7: "bad shape",
}


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("bp_oracle.c", "bp_oracle_impl.h", "qary_oracle.c", "mc_oracle.c")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def max_threads():
    return int(lib().oracle_max_threads())


METHODS = {"product_sum": 0, "ps": 0, "min_sum": 1, "ms": 1, "min_sum_log": 1, "msl": 1, "product_sum_log": 2, "psl": 2,
           "tanh_complement": 3}


def bp_decode_batch(g, channel_probs, inputs, mode, max_iter, method, alpha=1.0, dtype="f64", threads=1, early_exit=True):
    """g: TannerGraph-like (m, n, row_ptr, col_idx, col_ptr, csc_edge).
    inputs: uint8 [batch, m] (mode 0, syndromes) or [batch, n] (mode 1, received).
    Returns dict(bits uint8 [batch,n], llr [batch,n], iters int32, converged int32)."""
    L = lib()
    inputs = np.ascontiguousarray(inputs, dtype=np.uint8)
    if inputs.ndim == 1:
        inputs = inputs[None, :]
    batch = inputs.shape[0]
    want = g.n if mode else g.m
    if inputs.shape[1] != want:
        raise ValueError(f"input length {inputs.shape[1]} != {want}")
    probs = np.ascontiguousarray(channel_probs, dtype=np.float64)
    assert probs.shape == (g.n,)
    ft = np.float64 if dtype == "f64" else np.float32
    ct = C.c_double if dtype == "f64" else C.c_float
    bits = np.zeros((batch, g.n), dtype=np.uint8)
    llr = np.zeros((batch, g.n), dtype=ft)
    iters = np.zeros(batch, dtype=np.int32)
    conv = np.zeros(batch, dtype=np.int32)
    fn = getattr(L, f"oracle_bp_decode_batch_{dtype}")
    m = METHODS[method] if isinstance(method, str) else int(method)
    rc = fn(
        C.c_int(g.m), C.c_int(g.n), _p(g.row_ptr, C.c_int32), _p(g.col_idx, C.c_int32),
        _p(g.col_ptr, C.c_int32), _p(g.csc_edge, C.c_int32), _p(probs, C.c_double),
        _p(inputs, C.c_uint8), C.c_int(mode), C.c_int(batch), C.c_int(max_iter), C.c_int(m),
        C.c_double(alpha), _p(bits, C.c_uint8), _p(llr, ct), _p(iters, C.c_int32), _p(conv, C.c_int32),
        C.c_int(threads), C.c_int(1 if early_exit else 0),
    )  # fmt: skip
    if rc:
        raise RuntimeError(f"oracle_bp_decode_batch failed: {rc}")
    return {"bits": bits, "llr": llr, "iters": iters, "converged": conv}


def qary_into_llr(pmf):
    pmf = np.ascontiguousarray(pmf, dtype=np.float32)
    out = np.zeros_like(pmf)
    rc = lib().oracle_qary_into_llr(C.c_int(pmf.shape[0]), C.c_int(pmf.shape[1]), _p(pmf, C.c_float), _p(out, C.c_float))
    if rc:
        raise RuntimeError(QERR.get(rc, str(rc)))
    return out


def qary_min_sum_batch(g, Q, pmf, max_iter, threads=1):
    """pmf float32 [batch, N, Q] (or [N, Q]) -> int8 [batch, N] hard decisions."""
    pmf = np.ascontiguousarray(pmf, dtype=np.float32)
    single = pmf.ndim == 2
    if single:
        pmf = pmf[None]
    batch = pmf.shape[0]
    assert pmf.shape[1:] == (g.n, Q)
    out = np.zeros((batch, g.n), dtype=np.int8)
    rc = lib().oracle_qary_min_sum_batch(
        C.c_int(g.m), C.c_int(g.n), C.c_int(Q), _p(g.row_ptr, C.c_int32), _p(g.col_idx, C.c_int32),
        _p(g.val, C.c_int8), _p(g.col_ptr, C.c_int32), _p(g.csc_edge, C.c_int32), _p(pmf, C.c_float),
        C.c_int(batch), C.c_int(max_iter), _p(out, C.c_int8), C.c_int(threads),
    )  # fmt: skip
    if rc:
        raise RuntimeError(QERR.get(rc, str(rc)))
    return out[0] if single else out


def qary_special_batch(g, B, BSUM, pmf_b, pmf_s, max_iter, threads=1):
    pmf_b = np.ascontiguousarray(pmf_b, dtype=np.float32)
    pmf_s = np.ascontiguousarray(pmf_s, dtype=np.float32)
    single = pmf_b.ndim == 2
    if single:
        pmf_b, pmf_s = pmf_b[None], pmf_s[None]
    batch = pmf_b.shape[0]
    assert pmf_b.shape[1:] == (g.n - g.m, 2 * B + 1) and pmf_s.shape[1:] == (g.m, 2 * BSUM + 1)
    out = np.zeros((batch, g.n), dtype=np.int8)
    rc = lib().oracle_qary_special_batch(
        C.c_int(g.m), C.c_int(g.n), C.c_int(B), C.c_int(BSUM), _p(g.row_ptr, C.c_int32),
        _p(g.col_idx, C.c_int32), _p(g.val, C.c_int8), _p(g.col_ptr, C.c_int32), _p(g.csc_edge, C.c_int32),
        _p(pmf_b, C.c_float), _p(pmf_s, C.c_float), C.c_int(batch), C.c_int(max_iter), _p(out, C.c_int8),
        C.c_int(threads),
    )  # fmt: skip
    if rc:
        raise RuntimeError(QERR.get(rc, str(rc)))
    return out[0] if single else out


def philox4x32_10(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().oracle_philox4x32_10(c, k, o)
    return [int(x) for x in o]


def mc_bernoulli(seed, first, batch, length, probs=None, p0=0.0):
    out = np.zeros((batch, length), dtype=np.uint8)
    pr = None if probs is None else np.ascontiguousarray(probs, dtype=np.float64)
    lib().oracle_mc_bernoulli(C.c_uint64(seed), C.c_int64(first), C.c_int(batch), C.c_int(length),
                              None if pr is None else _p(pr, C.c_double), C.c_double(p0), _p(out, C.c_uint8))
    return out


def mc_hqc_secret(seed, first, batch, N, omega):
    y = np.zeros((batch, omega), dtype=np.int32)
    lib().oracle_mc_hqc_secret(C.c_uint64(seed), C.c_int64(first), C.c_int(batch), C.c_int(N), C.c_int(omega), _p(y, C.c_int32))
    return y
