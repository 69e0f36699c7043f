"""ctypes binding of libscaldpc.so (the C ABI declared in include/scaldpc.h).

The product path has NO CPU fallback: if the HIP library is missing or cannot be
loaded, `load()` raises, and so does every decoder built on it.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
# (SCALDPC_SO: another build of the same library, for A/B measurements of compile-time variants)
SO_PATH = os.environ.get("SCALDPC_SO") or os.path.join(_HERE, "libscaldpc.so")
CSRC = os.path.join(_HERE, "csrc")

OK, EINVAL, EHIP, ENOMEM, EPMF, ENOCONF, EDEGREE = range(7)
BP_PRODUCT_SUM, BP_MIN_SUM = 0, 1
IN_SYNDROME, IN_RECEIVED = 0, 1
F_EARLY_EXIT, F_DEVICE_IO, F_ASYNC = 1, 2, 4

_lib = None


class ScaldpcError(RuntimeError):
    pass


def build(verbose=False):
    """Compile every HIP source for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(["make", "-C", CSRC, "../libscaldpc.so"], stdout=out)
    return SO_PATH


def _declare(lib):
    p = C.POINTER
    vp = C.c_void_p
    lib.scaldpc_last_error.restype = C.c_char_p
    lib.scaldpc_last_error.argtypes = []
    lib.scaldpc_version.restype = C.c_int
    lib.scaldpc_device_count.argtypes = [p(C.c_int)]
    lib.scaldpc_set_device.argtypes = [C.c_int]
    lib.scaldpc_trim.argtypes = []
    lib.scaldpc_trim.restype = C.c_int
    lib.scaldpc_debug_live_blocks.argtypes = [p(C.c_int64)]
    lib.scaldpc_debug_live_blocks.restype = C.c_int
    lib.scaldpc_debug_fail_alloc.argtypes = [C.c_int32]
    lib.scaldpc_debug_fail_alloc.restype = C.c_int
    lib.scaldpc_measure_rmw_stream.argtypes = [C.c_int64, C.c_int32, C.c_int32, p(C.c_double)]
    lib.scaldpc_measure_rmw_stream.restype = C.c_int
    lib.scaldpc_bp_create.argtypes = [C.c_int32, C.c_int32, C.c_int64, vp, vp, p(vp)]
    lib.scaldpc_bp_set_channel_probs.argtypes = [vp, vp]
    lib.scaldpc_bp_decode_batch.argtypes = [
        vp, vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_uint32, vp, vp, vp, vp, vp,
    ]  # fmt: skip
    lib.scaldpc_bp_time_kernels.argtypes = [vp, C.c_int32, C.c_int32, C.c_float, vp, p(C.c_float), p(C.c_int32)]
    lib.scaldpc_bp_set_tile_group.argtypes = [vp, C.c_int32]
    lib.scaldpc_bp_last_compacted.argtypes = [vp, p(C.c_int64)]
    lib.scaldpc_bp_last_compacted.restype = C.c_int
    lib.scaldpc_bp_last_stats.argtypes = [vp, p(C.c_int64)]
    lib.scaldpc_bp_last_stats.restype = C.c_int
    lib.scaldpc_bp_append_rows.argtypes = [vp, C.c_int32, vp, vp, C.c_int32]
    lib.scaldpc_bp_append_rows.restype = C.c_int
    lib.scaldpc_bp_set_channel_probs_tail.argtypes = [vp, C.c_int32, C.c_int32, vp]
    lib.scaldpc_bp_set_channel_probs_tail.restype = C.c_int
    lib.scaldpc_bp_configure.argtypes = [vp, C.c_char_p, C.c_char_p]
    lib.scaldpc_bp_configure.restype = C.c_int
    lib.scaldpc_bp_device_of.argtypes = [vp, p(C.c_int32)]
    lib.scaldpc_bp_device_of.restype = C.c_int
    lib.scaldpc_bp_destroy.argtypes = [vp]
    lib.scaldpc_bp_destroy.restype = None
    lib.scaldpc_mc_fer_run.argtypes = [
        vp, C.c_int64, C.c_int32, C.c_uint64, C.c_int32, C.c_int32, C.c_float, C.c_uint32, vp, vp, vp, vp,
    ]  # fmt: skip
    lib.scaldpc_mc_hqc_run.argtypes = [
        vp, C.c_int32, C.c_double, C.c_int64, C.c_int32, C.c_uint64, C.c_int32, C.c_int32, C.c_float, C.c_uint32,
        vp, vp, vp, vp, vp,
    ]  # fmt: skip
    for name in (
        "scaldpc_device_count", "scaldpc_set_device", "scaldpc_bp_create", "scaldpc_bp_set_channel_probs",
        "scaldpc_bp_decode_batch", "scaldpc_bp_time_kernels", "scaldpc_bp_set_tile_group", "scaldpc_mc_fer_run",
        "scaldpc_mc_hqc_run",
    ):  # fmt: skip
        getattr(lib, name).restype = C.c_int
    lib.scaldpc_qary_create.argtypes = [C.c_int32, C.c_int32, C.c_int32, vp, C.c_int32, p(vp)]
    lib.scaldpc_qary_min_sum_batch.argtypes = [vp, vp, C.c_int32, C.c_uint32, vp, vp]
    lib.scaldpc_qary_configure.argtypes = [vp, C.c_char_p, C.c_char_p]
    lib.scaldpc_qary_configure.restype = C.c_int
    lib.scaldpc_qary_last_timing.argtypes = [vp, p(C.c_float), p(C.c_int32)]
    lib.scaldpc_qary_last_timing.restype = C.c_int
    lib.scaldpc_qary_into_llr.argtypes = [vp, C.c_int64, C.c_int32, C.c_uint32, vp, vp]
    lib.scaldpc_qary_into_llr.restype = C.c_int
    lib.scaldpc_qary_destroy.argtypes = [vp]
    lib.scaldpc_qary_destroy.restype = None
    lib.scaldpc_qary_special_create.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, vp, C.c_int32, p(vp)]
    lib.scaldpc_qary_special_min_sum_batch.argtypes = [vp, vp, vp, C.c_int32, C.c_uint32, vp, vp]
    for name in (
        "scaldpc_qary_create", "scaldpc_qary_min_sum_batch", "scaldpc_qary_special_create",
        "scaldpc_qary_special_min_sum_batch",
    ):  # fmt: skip
        getattr(lib, name).restype = C.c_int


def load():
    """Load libscaldpc.so (once).  Raises if it is not built -- no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise ScaldpcError(
            f"{SO_PATH} is missing: build it with `make -C {CSRC}` (or __graft_entry__.build()). "
            "sca-ldpc_amd has no CPU fallback."
        )
    # PyTorch ships its own libamdhip64 under the same soname; when torch is part of
    # the process (bench, multi-GPU plumbing) it must be loaded first so that this
    # library binds to the same HIP runtime instead of a second copy.
    if "torch" not in sys.modules and os.environ.get("SCALDPC_NO_TORCH") != "1":
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    lib = C.CDLL(SO_PATH, mode=C.RTLD_GLOBAL)
    _declare(lib)
    _lib = lib
    return lib


def check(rc):
    if rc == OK:
        return
    msg = load().scaldpc_last_error().decode(errors="replace")
    if rc == EINVAL:
        raise ValueError(msg)
    if rc == ENOMEM:
        raise MemoryError(msg)
    raise ScaldpcError(f"[{rc}] {msg}")


def trim():
    """Return the device / pinned blocks parked by destroyed decoders to the driver."""
    check(load().scaldpc_trim())


def live_blocks():
    """Allocator accounting: blocks live handles own (device / pinned host) and blocks parked for reuse."""
    out = (C.c_int64 * 6)()
    check(load().scaldpc_debug_live_blocks(out))
    keys = ("device_blocks", "device_bytes", "pinned_blocks", "pinned_bytes", "idle_blocks", "idle_bytes")
    return dict(zip(keys, (int(v) for v in out)))


def measure_rmw_stream(nbytes, rows_per_wave=51, reps=50):
    """GB/s (read + written) of an in-place read-all / write-all stream over `nbytes` of scratch device memory, every
    wave sweeping `rows_per_wave` consecutive 256-B rows: the access shape of an in-place BP pass (bench.py's
    cache / HBM ceilings; a measurement aid, nothing the decode path calls)."""
    out = C.c_double(0.0)
    check(load().scaldpc_measure_rmw_stream(int(nbytes), int(rows_per_wave), int(reps), C.byref(out)))
    return float(out.value)


def ptr(a):
    """Host numpy array or device pointer (int) -> c_void_p."""
    if a is None:
        return None
    if isinstance(a, int):
        return C.c_void_p(a)
    return C.c_void_p(a.ctypes.data)
