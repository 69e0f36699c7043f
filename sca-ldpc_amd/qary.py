"""`simulate_rs`-shaped front end of the HIP q-ary min-sum decoders.

The reference generates one PyO3 class per compile-time size
(`register_py_decoder_class!`, simulate_rs/src/pydecoder.rs:12-70; sizes listed in
simulate_rs/src/lib.rs:32-75) and looks them up by name
(`getattr(simulate_rs, f"DecoderN{n}R{r}V{v}C{c}B{B}")`, simulate/decode.py:227-229).
Here sizes are run-time values: `decoder_class("DecoderN450R150V3C7B1")` builds a class
with the same constructor / `min_sum` surface for ANY name of that pattern, plus
`min_sum_batch` for many channel outputs per call.

  DecoderN{N}R{R}V{DV}C{DC}B{B}(H: int8 [R, N], iterations)   .min_sum(pmf float32 [N, 2B+1]) -> list[int]
  DecoderN{N}R{R}SW{SW}(H: int8 [R, N], iterations)           .min_sum(pmf [N-R, 5], pmf_sum [R, 2*BSUM+1]) -> list[int]
      (B = 2, BSUM = SW*B, DC = SW+1: the Kyber decoders of lib.rs:54-75)

Inputs are probabilities; the LLR conversion (decoder.rs:668-692) happens inside, as in
the reference.  Errors: a pmf row not summing to 1 +- 1e-3 and a check without any
finite configuration raise (the reference panics); shape mismatches raise ValueError.
`min_sum` may be called concurrently from many Python threads on one object (the
reference's thread pool does, decode.py:247-262): calls are serialised in the library
and ctypes releases the GIL meanwhile.
"""
from __future__ import annotations

import ctypes as C
import re

import numpy as np

from . import _lib

_GENERIC = re.compile(r"^DecoderN(\d+)R(\d+)V(\d+)C(\d+)B(\d+)$")
_SPECIAL = re.compile(r"^DecoderN(\d+)R(\d+)SW(\d+)$")


class _QaryBase:
    N = R = DV = DC = B = Q = 0

    def _check_H(self, H):
        H = np.asarray(H)
        if H.dtype != np.int8:
            raise TypeError("parity check matrix must have dtype int8 (as PyReadonlyArray2<i8>, pydecoder.rs:24)")
        if H.shape != (self.R, self.N):
            raise ValueError(f"parity check matrix has shape {H.shape}, this decoder is built for ({self.R}, {self.N})")
        nz = H != 0
        if nz.sum(axis=0).max() > self.DV or nz.sum(axis=1).max() > self.DC:
            # the reference panics in insert_first_none (decoder.rs:465-473)
            raise ValueError("Reached the end of the array, no more space left! (node degree exceeds DV/DC)")
        return np.ascontiguousarray(H)

    def configure(self, **knobs):
        """wave = -1 (auto) / 0 / 1, unroll = 0 / 1, tree = 0 / 1, dp = 0 / 1, dp_min = batch from which dp applies, timing = 0 / 1
        (include/scaldpc.h, scaldpc_qary_configure)."""
        for k, v in knobs.items():
            _lib.check(self._lib.scaldpc_qary_configure(self._h, k.encode(), str(v).encode()))

    CHECK_KERNELS = ("k_q_check_unrolled<3,7>", "k_q_check_unrolled<5,5>", "k_q_special_check_tree<5,6>",
                     "k_q_special_check_wave", "k_q_check_wave", "k_q_special_check", "k_q_check", "k_q_special_check_dp<5,6>", "k_q_check_dp<3,7>")

    def last_timing(self):
        """HIP-event times of the last call's launches (after `configure(timing=1)`; bench.py's measurement aid):
        dict(ms_check, ms_var, ms_loop, iterations, check_kernel, batch, max_check_degree)."""
        ms = (C.c_float * 3)()
        info = (C.c_int32 * 4)()
        _lib.check(self._lib.scaldpc_qary_last_timing(self._h, ms, info))
        return {"ms_check": ms[0], "ms_var": ms[1], "ms_loop": ms[2], "iterations": info[0],
                "check_kernel": self.CHECK_KERNELS[info[1]] if info[1] >= 0 else None, "batch": info[2],
                "max_check_degree": info[3]}

    def close(self):
        if getattr(self, "_h", None):
            self._lib.scaldpc_qary_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class QaryDecoder(_QaryBase):
    """Decoder<N, R, DV, DC, Q=2B+1, B, i8> (decoder.rs:417-438)."""

    def __init__(self, py_parity_check, iterations):
        self._h = None
        self._lib = _lib.load()
        H = self._check_H(py_parity_check)
        h = C.c_void_p()
        _lib.check(self._lib.scaldpc_qary_create(self.R, self.N, self.B, _lib.ptr(H), int(iterations), C.byref(h)))
        self._h = h

    def min_sum_batch(self, channel_output):
        """float32 [batch, N, Q] -> int8 [batch, N]."""
        p = np.ascontiguousarray(channel_output, dtype=np.float32)
        if p.ndim != 3 or p.shape[1:] != (self.N, self.Q):
            raise ValueError(f"channel output has shape {p.shape}, expected (batch, {self.N}, {self.Q})")
        out = np.empty((p.shape[0], self.N), dtype=np.int8)
        _lib.check(self._lib.scaldpc_qary_min_sum_batch(self._h, _lib.ptr(p), p.shape[0], 0, None, _lib.ptr(out)))
        return out

    def min_sum_batch_device(self, d_channel_output, batch, d_out, stream=0):
        """Device-pointer variant (ints, e.g. torch `tensor.data_ptr()`): float32 [batch, N, Q] probabilities in HBM ->
        int8 [batch, N] in HBM; nothing crosses PCIe.  Returns after the stream work is complete."""
        _lib.check(self._lib.scaldpc_qary_min_sum_batch(self._h, C.c_void_p(d_channel_output), int(batch), _lib.F_DEVICE_IO,
                                                        C.c_void_p(stream or None), C.c_void_p(d_out)))

    def min_sum(self, py_channel_output):
        p = np.asarray(py_channel_output)
        if p.shape != (self.N, self.Q):
            raise ValueError(f"channel output has shape {p.shape}, expected ({self.N}, {self.Q})")
        return [int(x) for x in self.min_sum_batch(p[None])[0]]


class QarySpecialDecoder(_QaryBase):
    """DecoderSpecial<N, R, N-R, DC-1, DC, DV, B, 2B+1, BSUM, 2BSUM+1, i8> (decoder_special.rs:294-322)."""

    BSUM = QS = 0

    def __init__(self, py_parity_check, iterations):
        self._h = None
        self._lib = _lib.load()
        H = self._check_H(py_parity_check)
        h = C.c_void_p()
        _lib.check(
            self._lib.scaldpc_qary_special_create(self.R, self.N, self.B, self.BSUM, _lib.ptr(H), int(iterations), C.byref(h))
        )
        self._h = h

    def min_sum_batch(self, channel_output, channel_output_sum):
        p = np.ascontiguousarray(channel_output, dtype=np.float32)
        ps = np.ascontiguousarray(channel_output_sum, dtype=np.float32)
        if p.ndim != 3 or p.shape[1:] != (self.N - self.R, self.Q):
            raise ValueError(f"channel output has shape {p.shape}, expected (batch, {self.N - self.R}, {self.Q})")
        if ps.shape != (p.shape[0], self.R, self.QS):
            raise ValueError(f"channel output sum has shape {ps.shape}, expected ({p.shape[0]}, {self.R}, {self.QS})")
        out = np.empty((p.shape[0], self.N), dtype=np.int8)
        _lib.check(
            self._lib.scaldpc_qary_special_min_sum_batch(self._h, _lib.ptr(p), _lib.ptr(ps), p.shape[0], 0, None, _lib.ptr(out))
        )
        return out

    def min_sum_batch_device(self, d_channel_output, d_channel_output_sum, batch, d_out, stream=0):
        """Device-pointer variant: float32 [batch, N-R, 2B+1] and [batch, R, 2BSUM+1] in HBM -> int8 [batch, N] in HBM."""
        _lib.check(self._lib.scaldpc_qary_special_min_sum_batch(self._h, C.c_void_p(d_channel_output), C.c_void_p(d_channel_output_sum),
                                                                int(batch), _lib.F_DEVICE_IO, C.c_void_p(stream or None),
                                                                C.c_void_p(d_out)))

    def min_sum(self, py_channel_output, py_channel_output_sum):
        p, ps = np.asarray(py_channel_output), np.asarray(py_channel_output_sum)
        if p.ndim != 2 or ps.ndim != 2:
            raise ValueError("channel outputs must be 2-D")
        return [int(x) for x in self.min_sum_batch(p[None], ps[None])[0]]


def into_llr(channel_output):
    """Decoder::into_llr (decoder.rs:668-692) on the device: float32 [rows, Q] probabilities ->
    float32 [rows, Q] LLRs ln(max / p), bit-identical to the host's logf; raises if a row does not
    sum to 1 +- 1e-3."""
    p = np.ascontiguousarray(channel_output, dtype=np.float32)
    if p.ndim != 2:
        raise ValueError("channel output must be 2-D [rows, Q]")
    out = np.empty_like(p)
    lib = _lib.load()
    _lib.check(lib.scaldpc_qary_into_llr(_lib.ptr(p), p.shape[0], p.shape[1], 0, None, _lib.ptr(out)))
    return out


# the enumeration kernels pack one 8-bit digit per edge of a check into a register word: 64 bits in every kernel
# (degree <= 8: all sizes the reference registers), 128 bits in the lane-per-codeword kernel of the plain decoder (<= 16)
MAX_CHECK_DEGREE = 16
MAX_SPECIAL_CHECK_DEGREE = 8


def _check_limits(name, DC, B, BSUM, special=False):
    """The reference generates a class per registered size (lib.rs:32-75: DC = 4, 7, 7, 7) and a name it
    has not registered is an AttributeError on `getattr(simulate_rs, name)` (decode.py:227-229).  Here
    any size resolves -- up to what the kernels are built for; beyond that the name does not resolve
    either, with the reason, at look-up time rather than at the first construction."""
    lim = MAX_SPECIAL_CHECK_DEGREE if special else MAX_CHECK_DEGREE
    if DC > lim:
        raise AttributeError(f"{name}: check degree {DC} > {lim} is not supported by the enumeration kernels "
                             f"(scaldpc_qary.hip; the reference's registered sizes use 4 and 7)")
    if B > 127 or BSUM > 127:
        raise AttributeError(f"{name}: symbols beyond +-127 do not fit the int8 hard decisions (decoder.rs i8)")


_cache = {}


def decoder_class(name: str):
    """Class for a `simulate_rs` decoder name (any size of either pattern)."""
    if name in _cache:
        return _cache[name]
    m = _GENERIC.match(name)
    if m:
        N, R, DV, DC, B = map(int, m.groups())
        _check_limits(name, DC, B, B)
        cls = type(name, (QaryDecoder,), dict(N=N, R=R, DV=DV, DC=DC, B=B, Q=2 * B + 1))
    else:
        m = _SPECIAL.match(name)
        if not m:
            raise AttributeError(name)
        N, R, SW = map(int, m.groups())
        B = 2  # Kyber eta (lib.rs:54-75: B = 2, BSUM = SW * B)
        _check_limits(name, SW + 1, B, SW * B, special=True)
        cls = type(
            name, (QarySpecialDecoder,),
            dict(N=N, R=R, DV=R, DC=SW + 1, B=B, Q=2 * B + 1, BSUM=SW * B, QS=2 * SW * B + 1),
        )  # fmt: skip
    _cache[name] = cls
    return cls
