"""`ldpc.bp_decoder`-shaped front end of the HIP binary BP decoder.

Mirrors the constructor/`decode()` surface the reference uses
(simulate/decode.py:155-161,171; simulate/hqc.py:694-699,708): same argument
names and meaning, same error behaviour (ValueError on a bad method, a bad
`channel_probs` length or a wrong-length input), same read-only result
attributes (`bp_decoding`, `log_prob_ratios`, `converge`, `iter`).  On top of
that: `decode_batch`, which decodes many independent inputs per call -- the
Monte-Carlo drivers' trials are independent (decode.py:165), so the batch
dimension is what feeds the GPU.

Differences from the reference package, all deliberate:
  * messages are fp32 (the package computes in float64);
  * "product_sum" runs the tanh rule in the LLR domain (the package's ratio-domain
    recursion is the same function of the messages);
  * H may be a dense array (as the reference passes), a scipy sparse matrix or a
    `TannerGraph`; it is never densified.
There is no CPU fallback: constructing a decoder without libscaldpc.so raises.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .graph import TannerGraph

_PS = {"prod_sum", "product_sum", "ps", "0", "prod sum", "prod_sum_log", "product_sum_log", "ps_log", "2", "psl"}
_MS = {"min_sum", "minimum_sum", "ms", "1", "minimum sum", "min sum", "min_sum_log", "minimum_sum_log", "ms_log", "3",
       "minimum sum_log", "msl"}  # fmt: skip


def _method_id(bp_method):
    key = str(bp_method).lower()
    if key in _PS:
        return _lib.BP_PRODUCT_SUM
    if key in _MS:
        return _lib.BP_MIN_SUM
    raise ValueError(
        f"BP method '{bp_method}' is invalid. Please choose from the following methods: "
        "'product_sum', 'minimum_sum', 'product_sum_log' or 'minimum_sum_log'"
    )


def _vector_type(t):
    if t in (-1, "auto", None):
        return -1
    if t in (0, "syndrome"):
        return _lib.IN_SYNDROME
    if t in (1, "received_vector", "received"):
        return _lib.IN_RECEIVED
    raise ValueError(f"input_vector_type '{t}' is invalid. Choose 'syndrome', 'received_vector' or 'auto'.")


class BpDecoder:
    def __init__(
        self,
        parity_check_matrix,
        error_rate=None,
        max_iter=0,
        bp_method=0,
        ms_scaling_factor=1.0,
        channel_probs=[None],
        input_vector_type=-1,
    ):
        self._h = None
        self._lib = _lib.load()  # raises when the HIP extension is missing
        g = TannerGraph.coerce(parity_check_matrix)
        self.graph = g
        self.m, self.n = g.m, g.n
        self.max_iter = int(max_iter) if int(max_iter) != 0 else self.n
        if self.max_iter < 0:
            raise ValueError("max_iter must be non-negative")
        self._method = _method_id(bp_method)
        self.bp_method = "product_sum" if self._method == _lib.BP_PRODUCT_SUM else "minimum_sum"
        self.ms_scaling_factor = float(ms_scaling_factor)
        self._vector_type = _vector_type(input_vector_type)
        h = C.c_void_p()
        _lib.check(
            self._lib.scaldpc_bp_create(g.m, g.n, g.nnz, _lib.ptr(g.row_ptr), _lib.ptr(g.col_idx), C.byref(h))
        )
        self._h = h
        # priors, as the package resolves them: channel_probs wins over error_rate
        # (no list() round trip: a 20 000-entry ndarray per decoder is the attack loop's normal case)
        have_cp = channel_probs is not None and len(channel_probs) > 0 and channel_probs[0] is not None
        if have_cp:
            if len(channel_probs) != self.n:
                raise ValueError(
                    f"The length of the channel probability vector must be eqaul to the block length n={self.n}."
                )
            probs = np.asarray(channel_probs, dtype=np.float64)
        elif error_rate is not None:
            probs = np.full(self.n, float(error_rate), dtype=np.float64)
        else:
            raise ValueError(
                "Please specify the error channel. Either: 1) error_rate: float or 2) channel_probs: list of floats "
                "indicating the error probability on each bit. "
            )
        self.update_channel_probs(probs)
        self.error_rate = error_rate
        # result attributes of the last decode()
        self.bp_decoding = np.zeros(self.n, dtype=int)
        self.log_prob_ratios = np.zeros(self.n, dtype=np.float64)
        self.converge = 0
        self.iter = 0

    # -- ldpc-compatible API --------------------------------------------------
    def update_channel_probs(self, channel):
        probs = np.ascontiguousarray(channel, dtype=np.float64)
        if probs.shape != (self.n,):
            raise ValueError(
                f"The length of the channel probability vector must be eqaul to the block length n={self.n}."
            )
        _lib.check(self._lib.scaldpc_bp_set_channel_probs(self._h, _lib.ptr(probs)))
        self._probs = probs
        self._probs_tails = []

    @property
    def channel_probs(self):
        if self._probs_tails:
            self._probs = np.concatenate([self._probs] + self._probs_tails)
            self._probs_tails = []
        return self._probs

    def append_rows(self, row_ptr, col_idx, new_n, channel_probs_tail):
        """The graph GROWS (the attack loop's `H = np.vstack([H, row])`, simulate/hqc.py:885-908, where the
        reference builds a new decoder per decode): append checks to this live decoder.

          row_ptr, col_idx    CSR of the NEW rows only (row_ptr[0] = 0; columns strictly ascending, < new_n)
          new_n               block length afterwards (>= n: new columns come last -- for H = [Hin | I] every
                              appended row brings its identity column)
          channel_probs_tail  priors of the new columns [n, new_n)

        Results of later decodes are those of a decoder freshly built on the grown graph, bit for bit."""
        rp = np.ascontiguousarray(row_ptr, dtype=np.int32)
        ci = np.ascontiguousarray(col_idx, dtype=np.int32)
        tail = np.ascontiguousarray(channel_probs_tail, dtype=np.float64)
        new_n = int(new_n)
        if rp.ndim != 1 or rp.size < 1 or ci.ndim != 1 or ci.size != int(rp[-1]):
            raise ValueError("append_rows expects CSR arrays of the new rows (row_ptr[0] = 0, len(col_idx) = row_ptr[-1])")
        if tail.shape != (new_n - self.n,):
            raise ValueError(f"channel_probs_tail must hold the priors of the {new_n - self.n} new columns")
        # validated BEFORE the graph grows: a bad prior must not leave a grown decoder with unset priors behind
        if tail.size and not bool(((tail >= 0.0) & (tail <= 1.0)).all()):
            bad = int(np.flatnonzero(~((tail >= 0.0) & (tail <= 1.0)))[0])
            raise ValueError(f"channel_probs[{self.n + bad}] = {tail[bad]} is not a probability")
        old_n = self.n
        _lib.check(self._lib.scaldpc_bp_append_rows(self._h, rp.size - 1, _lib.ptr(rp), _lib.ptr(ci), new_n))
        self.m += rp.size - 1
        self.n = new_n
        self.graph = None  # (the constructor's graph no longer describes this decoder)
        _lib.check(self._lib.scaldpc_bp_set_channel_probs_tail(self._h, old_n, new_n - old_n, _lib.ptr(tail)))
        self._probs_tails.append(tail)  # (`channel_probs` is put together when somebody asks: appends are on the attack loop's critical path)

    def _resolve_kind(self, length, input_vector_type=None):
        kind = self._vector_type if input_vector_type is None else _vector_type(input_vector_type)
        if kind == -1:
            if self.m == self.n:
                raise ValueError(
                    "parity check matrix is square: the input vector type is ambiguous, please set "
                    "input_vector_type to 'syndrome' or 'received_vector'"
                )
            if length == self.m:
                kind = _lib.IN_SYNDROME
            elif length == self.n:
                kind = _lib.IN_RECEIVED
        want = self.m if kind == _lib.IN_SYNDROME else self.n if kind == _lib.IN_RECEIVED else None
        if want is None or length != want:
            raise ValueError(
                f"The input to the ldpc.bp_decoder.decode must be either a received word (of length={self.n}) or a "
                f"syndrome (of length={self.m}). The inputted vector has length={length}. Valid formats are "
                "`np.ndarray` or `scipy.sparse.spmatrix`."
            )
        return kind

    def decode(self, input_vector):
        """One decode, reference semantics (early exit at H e == s).  Returns int array [n]."""
        if hasattr(input_vector, "toarray"):  # scipy.sparse vector, which the package accepts too
            input_vector = input_vector.toarray()
        v = np.asarray(input_vector)
        if v.ndim != 1:
            v = v.reshape(-1)
        r = self.decode_batch(v[None, :], want_llr=True)
        self.bp_decoding = r["bits"][0].astype(int)
        self.log_prob_ratios = r["llr"][0].astype(np.float64)
        self.converge = int(r["converged"][0])
        self.iter = int(r["iters"][0])
        return self.bp_decoding

    # -- batched API ------------------------------------------------------------
    def decode_batch(self, inputs, max_iter=None, early_exit=True, want_llr=False, input_vector_type=None):
        """inputs: [batch, m] syndromes or [batch, n] received words (any integer dtype).
        Returns dict(bits uint8 [batch, n], llr float32 [batch, n] or None,
        iters int32 [batch], converged uint8 [batch])."""
        x = np.asarray(inputs)
        if x.ndim != 2:
            raise ValueError("decode_batch expects a 2-D array [batch, length]")
        kind = self._resolve_kind(x.shape[1], input_vector_type)
        x = np.ascontiguousarray(x & 1 if x.dtype != np.uint8 else x, dtype=np.uint8)
        batch = x.shape[0]
        if batch == 0:
            raise ValueError("empty batch")
        bits = np.empty((batch, self.n), dtype=np.uint8)
        llr = np.empty((batch, self.n), dtype=np.float32) if want_llr else None
        iters = np.empty(batch, dtype=np.int32)
        conv = np.empty(batch, dtype=np.uint8)
        flags = _lib.F_EARLY_EXIT if early_exit else 0
        _lib.check(
            self._lib.scaldpc_bp_decode_batch(
                self._h, _lib.ptr(x), kind, batch, self.max_iter if max_iter is None else int(max_iter),
                self._method, self.ms_scaling_factor, flags, None, _lib.ptr(bits), _lib.ptr(llr),
                _lib.ptr(iters), _lib.ptr(conv),
            )  # fmt: skip
        )
        return {"bits": bits, "llr": llr, "iters": iters, "converged": conv}

    def decode_batch_device(self, d_in, kind, batch, d_out_bits, max_iter=None, early_exit=False, stream=0,
                            d_out_llr=0, d_out_iters=0, d_out_conv=0, asynchronous=False):
        """Device-pointer variant (ints, e.g. torch `tensor.data_ptr()`): nothing crosses PCIe."""
        flags = _lib.F_DEVICE_IO | (_lib.F_EARLY_EXIT if early_exit else 0) | (_lib.F_ASYNC if asynchronous else 0)
        _lib.check(
            self._lib.scaldpc_bp_decode_batch(
                self._h, C.c_void_p(d_in), kind, batch, self.max_iter if max_iter is None else int(max_iter),
                self._method, self.ms_scaling_factor, flags, C.c_void_p(stream or None), C.c_void_p(d_out_bits),
                C.c_void_p(d_out_llr or None), C.c_void_p(d_out_iters or None), C.c_void_p(d_out_conv or None),
            )  # fmt: skip
        )

    def time_kernels(self, iters, stream=0):
        """HIP-event timing of the check / variable kernels on the last decode's state.
        Returns dict(ms_check, ms_var, launches_check, launches_var, codewords per check launch, lanes);
        lanes = 2: the two series ran concurrently on the two streams of the decode's schedule."""
        ms = (C.c_float * 2)()
        ln = (C.c_int32 * 6)()
        _lib.check(
            self._lib.scaldpc_bp_time_kernels(
                self._h, iters, self._method, self.ms_scaling_factor, C.c_void_p(stream or None), ms, ln
            )
        )
        return {"ms_check": ms[0], "ms_var": ms[1], "launches_check": ln[0], "launches_var": ln[1], "codewords": ln[2],
                "lanes": ln[3], "codewords_var": ln[4], "record_form": bool(ln[5] & 1),
                # the variable pass is timed in the form the last decode launched it in (early exit: every pass writes
                # decisions for all columns; fixed iterations: no output, record form without the degree <= 1 columns)
                "var_writes_out": bool(ln[5] & 2), "var_slim": bool(ln[5] & 4)}

    # -- Monte-Carlo helpers on the device (K6) --------------------------------------
    def mc_fer_run(self, runs, seed, first_trial=0, max_iter=None, early_exit=True, want_errors=False):
        """`runs` trials of the FER loop body (simulate/decode.py:165-175) entirely on the
        device: error ~ Bernoulli(channel_probs), syndrome, decode, compare.
        Returns dict(success uint8 [runs], iters int32 [runs], errors uint8 [runs, n] or None)."""
        succ = np.empty(runs, dtype=np.uint8)
        iters = np.empty(runs, dtype=np.int32)
        err = np.empty((runs, self.n), dtype=np.uint8) if want_errors else None
        _lib.check(
            self._lib.scaldpc_mc_fer_run(
                self._h, int(first_trial), int(runs), int(seed), self.max_iter if max_iter is None else int(max_iter),
                self._method, self.ms_scaling_factor, _lib.F_EARLY_EXIT if early_exit else 0, None, _lib.ptr(succ),
                _lib.ptr(iters), _lib.ptr(err),
            )  # fmt: skip
        )
        return {"success": succ, "iters": iters, "errors": err}

    def mc_hqc_run(self, runs, omega, eps, seed, first_trial=0, max_iter=None, early_exit=True, want_inputs=False):
        """`runs` synthetic hqc.decode() trials (simulate/hqc.py:684-705,742-749) on H = [Hin | I]:
        secret y, noisy checks, msg = [0]*N ++ checks, decode, success = (decoded[:N] == y).
        Returns dict(success, iters, msg uint8 [runs, n] or None, y int32 [runs, omega] or None)."""
        succ = np.empty(runs, dtype=np.uint8)
        iters = np.empty(runs, dtype=np.int32)
        msg = np.empty((runs, self.n), dtype=np.uint8) if want_inputs else None
        y = np.empty((runs, omega), dtype=np.int32) if want_inputs else None
        _lib.check(
            self._lib.scaldpc_mc_hqc_run(
                self._h, int(omega), float(eps), int(first_trial), int(runs), int(seed),
                self.max_iter if max_iter is None else int(max_iter), self._method, self.ms_scaling_factor,
                _lib.F_EARLY_EXIT if early_exit else 0, None, _lib.ptr(succ), _lib.ptr(iters), _lib.ptr(msg), _lib.ptr(y),
            )  # fmt: skip
        )
        return {"success": succ, "iters": iters, "msg": msg, "y": y}

    def last_compacted(self):
        """Codewords the last early-exit call re-decoded in its compact second pass."""
        c = C.c_int64()
        _lib.check(self._lib.scaldpc_bp_last_compacted(self._h, C.byref(c)))
        return int(c.value)

    def last_stats(self):
        """Path statistics of the last call: codewords handed to the compact pass, codewords decoded
        with the row-parallel (lane = edge) kernels, deepest compaction level reached."""
        out = (C.c_int64 * 4)()
        _lib.check(self._lib.scaldpc_bp_last_stats(self._h, out))
        return {"compacted": int(out[0]), "row_parallel": int(out[1]), "levels": int(out[2])}

    def last_row_parallel(self):
        return self.last_stats()["row_parallel"]

    def configure(self, **knobs):
        """Tuning / test knobs of this decoder (include/scaldpc.h, scaldpc_bp_configure): path, split,
        group_mb, el_max, el_fuse, compact_after, var_order, first_fused, fuse_test, minsum_rec, rec_skip1.  A new decoder takes its
        defaults from the SCALDPC_* environment once, at construction; results never depend on them."""
        for k, v in knobs.items():
            _lib.check(self._lib.scaldpc_bp_configure(self._h, k.encode(), str(v).encode()))

    def device_of(self):
        """{device the decoder was created on, device of its graph / message / state allocations (-1: none yet)}."""
        out = (C.c_int32 * 4)()
        _lib.check(self._lib.scaldpc_bp_device_of(self._h, out))
        return {"device": out[0], "graph": out[1], "messages": out[2], "state": out[3]}

    def set_tile_group(self, tiles):
        _lib.check(self._lib.scaldpc_bp_set_tile_group(self._h, int(tiles)))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.scaldpc_bp_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


bp_decoder = BpDecoder
