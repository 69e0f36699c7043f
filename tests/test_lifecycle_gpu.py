"""Handle lifetime: what a decoder allocates, its destroy gives back -- device AND pinned host.

The reference builds a NEW decoder for every decode (simulate/hqc.py:694-708: dense H ->
`bp_decoder(...)` -> one `decode()`), so a block the destroy forgets is a leak per decode of
the zero-change drop-in route.  `scaldpc_debug_live_blocks` counts the blocks live handles
own; a create / decode / close cycle must leave that count where it found it.

The second half runs in a child process with SCALDPC_NO_CACHE=1 (released blocks really go
back to the driver: a stale pointer is a use-after-free, not a parked block) and
SCALDPC_POISON=1 (every block handed out is 0xFF-filled): one live handle is called with a
small, a larger and again the small `max_iter` / batch, and must give the oracle's answers
each time."""
import importlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import S, hqc_instance

pytestmark = pytest.mark.gpu
bp = importlib.import_module("sca-ldpc_amd.bp")
qary = importlib.import_module("sca-ldpc_amd.qary")
lib = importlib.import_module("sca-ldpc_amd._lib")
trials = importlib.import_module("sca-ldpc_amd.trials")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIVE = ("device_blocks", "device_bytes", "pinned_blocks", "pinned_bytes")


def live():
    b = lib.live_blocks()
    return tuple(b[k] for k in LIVE)


def test_decoder_per_decode_leaves_no_block_behind():
    """200 x {build a decoder on the HQC-128 graph, one host-buffer decode() with posteriors, close}
    -- hqc.py:694-708's pattern -- returns every device and pinned block."""
    rows = json.load(open(os.path.join(ROOT, "tests", "golden", "hqc_first_rows.json")))
    H, Hin, _ = S.codes.hqc_bench_graph("hqc128", rows["N17669_W50_s0"], R=1500)
    N, omega = S.codes.HQC_PARAMS["hqc128"]
    probs = trials.hqc_priors(N, Hin.m, omega, 0.02)
    msg, ys = trials.hqc_trials(Hin, omega, 0.02, 4, base_seed=2, first_index=0)
    base = live()
    first = None
    for k in range(200):
        dec = bp.bp_decoder(H, max_iter=100, bp_method="product_sum", channel_probs=probs)
        out = dec.decode(msg[k % 4])
        if k == 0:
            during = live()
            assert during[0] > base[0] and during[2] > base[2], "the decode allocated nothing?"
            first = (out.copy(), dec.log_prob_ratios.copy(), dec.iter)
        elif k % 4 == 0:
            assert np.array_equal(out, first[0]) and np.array_equal(dec.log_prob_ratios, first[1]) and dec.iter == first[2]
        dec.close()
        assert live() == base, f"cycle {k}: live blocks {live()} != {base}"
    # the other entry points of a handle: batched tiles with early exit (compaction levels), Monte-Carlo, append
    dec = bp.bp_decoder(H, max_iter=40, bp_method="min_sum", channel_probs=probs)
    dec.configure(path="stream")
    big, _ = trials.hqc_trials(Hin, omega, 0.02, 300, base_seed=3, first_index=0)
    dec.decode_batch(big, early_exit=True, want_llr=True)
    dec.mc_hqc_run(200, omega=omega, eps=0.02, seed=1)
    dec.mc_fer_run(100, seed=2)
    dec.close()
    assert live() == base
    half = 700
    Hh = S.codes.hqc_bench_graph("hqc128", rows["N17669_W50_s0"], R=half)[0]
    dec = bp.bp_decoder(Hh, max_iter=30, bp_method="product_sum", channel_probs=np.concatenate([probs[:N], probs[N:N + half]]))
    dec.decode(np.concatenate([msg[0, :N], msg[0, N:N + half]]))
    rp = H.row_ptr[half:H.m + 1].astype(np.int64)
    dec.append_rows((rp - rp[0]).astype(np.int32), H.col_idx[rp[0]:rp[-1]], H.n, probs[N + half:])
    dec.decode(msg[0])
    dec.close()
    assert live() == base


def test_qary_decoder_per_call_leaves_no_block_behind(golden):
    g = S.TannerGraph.from_coo(golden["generators"]["regular_identity_300_150_3_6_s1"])
    Hd = g.to_dense(np.int8)
    cls = qary.decoder_class("DecoderN450R150V3C7B1")
    rng = np.random.RandomState(3)
    pmf = rng.dirichlet(np.ones(3) * 4, size=(5, 450)).astype(np.float32)
    base = live()
    for k in range(50):
        d = cls(Hd, 5)
        d.min_sum(pmf[k % 5])
        if k % 10 == 0:
            d.min_sum_batch(np.repeat(pmf, 40, axis=0))  # a larger batch re-makes the workspaces
        d.close()
        assert live() == base, f"cycle {k}"
    qary.into_llr(pmf[0])
    assert live() == base


CHILD = r"""
import importlib, json, os, sys
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
from helpers import ORACLE_METHOD, S, compare, hqc_instance
from oracle import pyoracle
bp = importlib.import_module("sca-ldpc_amd.bp"); qary = importlib.import_module("sca-ldpc_amd.qary")
lib = importlib.import_module("sca-ldpc_amd._lib")
base = lib.live_blocks()
H, Hin, probs, msg, y = hqc_instance(997, 9, 300, 6, 0.04, 5, seed=31)
other = bp.bp_decoder(H, max_iter=10, bp_method="min_sum", channel_probs=probs)   # a second live handle: a parked
for method in ("min_sum", "product_sum"):                                         # staging block would land in it
    for path, nb in (("edge", 3), ("auto", 5), ("stream", 5)):
        dec = bp.bp_decoder(H, max_iter=20, bp_method=method, channel_probs=probs)
        dec.configure(path=path)
        for mi in (20, 100, 20, 150, 7):
            got = dec.decode_batch(msg[:nb], max_iter=mi, want_llr=True)
            other.decode_batch(msg[:2], max_iter=mi + 1, want_llr=True)
            ref = pyoracle.bp_decode_batch(H, probs, msg[:nb], 1, mi, ORACLE_METHOD[method], dtype="f32", threads=4)
            compare(got, ref, method)
        dec.close()
other.close()
g = S.TannerGraph.from_coo(json.load(open(os.path.join(sys.argv[1], "tests", "golden", "generators.json")))["regular_identity_300_150_3_6_s1"])
Hd = g.to_dense(np.int8)
d = qary.decoder_class("DecoderN450R150V3C7B1")(Hd, 5)
rng = np.random.RandomState(3)
pmf = rng.dirichlet(np.ones(3) * 4, size=(130, 450)).astype(np.float32)
for nb in (1, 130, 3, 70):
    got = d.min_sum_batch(pmf[:nb])
    ref = pyoracle.qary_min_sum_batch(g, 3, pmf[:nb], 5, threads=4)
    assert np.array_equal(got, ref), nb
d.close()
end = lib.live_blocks()
assert all(end[k] == base[k] for k in ("device_blocks", "device_bytes", "pinned_blocks", "pinned_bytes")), (base, end)
assert end["idle_blocks"] == 0, "SCALDPC_NO_CACHE=1 must park nothing"
print("child ok")
"""


def test_one_handle_with_growing_and_shrinking_calls_without_the_block_cache():
    env = dict(os.environ, SCALDPC_NO_CACHE="1", SCALDPC_POISON="1")
    env.pop("SCALDPC_PATH", None)
    r = subprocess.run([sys.executable, "-c", CHILD, ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "child ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.fixture
def fail_alloc():
    """The fault injector, disarmed again when the test ends -- however it ends: an armed countdown is process-wide."""
    L = lib.load()
    assert L.scaldpc_debug_fail_alloc(0) == 0
    yield L.scaldpc_debug_fail_alloc
    L.scaldpc_debug_fail_alloc(0)


def test_failed_append_leaves_a_refusing_handle_and_no_leak(fail_alloc):
    """scaldpc_bp_append_rows with the k-th allocation failing (scaldpc_debug_fail_alloc), k = 1, 2, ...: the call
    returns an error code (MemoryError here), the handle then REFUSES further calls -- host mirrors and device arrays
    may disagree, decoding on them would be undefined -- and destroy still releases every block.  Once k is past the
    allocations an append makes, the append succeeds and the decoder gives the fresh decoder's answers.  Both kinds
    of append are injected into: the first one of a handle (its arrays move to growable allocations) and a later one
    (row-parallel tables updated in place)."""
    L = lib.load()
    H, Hin, probs, msg, y = hqc_instance(997, 9, 300, 6, 0.03, 3, seed=41)
    N = 997

    def part(r):  # the first r rows of [Hin | I]: graph, priors, inputs
        g = S.TannerGraph.from_csr(r, N + r, H.row_ptr[: r + 1].copy(), H.col_idx[: H.row_ptr[r]].copy())
        return g, np.concatenate([probs[:N], probs[N:N + r]]), np.concatenate([msg[:, :N], msg[:, N:N + r]], axis=1)

    def rows(r0, r1):  # CSR of rows [r0, r1), the block length afterwards, the priors of their identity columns
        rp = H.row_ptr[r0:r1 + 1].astype(np.int64)
        return (rp - rp[0]).astype(np.int32), H.col_idx[rp[0]:rp[-1]], N + r1, probs[N + r0:N + r1]

    fresh = bp.bp_decoder(H, max_iter=30, bp_method="min_sum", channel_probs=probs)
    fresh.configure(path="edge")
    want = fresh.decode_batch(msg, want_llr=True)
    fresh.close()
    base = live()
    for first_append in (True, False):
        failed = succeeded = 0
        for k in range(1, 40):
            g0, p0, x0 = part(100)
            dec = bp.bp_decoder(g0, max_iter=30, bp_method="min_sum", channel_probs=p0)
            dec.configure(path="edge")
            dec.decode_batch(x0)
            r0 = 100
            if not first_append:
                dec.append_rows(*rows(100, 150))
                dec.decode_batch(part(150)[2])  # the row-parallel tables of the growable handle exist now
                r0 = 150
            assert fail_alloc(k) == 0  # (armed: SCALDPC_DEBUG=1, tests/conftest.py)
            try:
                dec.append_rows(*rows(r0, 300))
                fail_alloc(0)
                got = dec.decode_batch(msg, want_llr=True)
                for key in ("bits", "llr", "iters", "converged"):
                    assert np.array_equal(got[key], want[key]), (k, key)
                succeeded += 1
            except MemoryError:
                fail_alloc(0)
                failed += 1
                with pytest.raises(Exception, match="unusable"):
                    dec.decode_batch(part(r0)[2])  # (the Python object still has the old block length)
                with pytest.raises(Exception, match="unusable"):
                    dec.append_rows(*rows(r0, 300))
            dec.close()
            assert live() == base, f"k = {k}: {live()} != {base}"
            if succeeded >= 2:
                break
        assert failed >= 1 and succeeded >= 2, (first_append, failed, succeeded)


def test_accumulator_survives_a_bad_certainty():
    """driver.HqcCheckAccumulator keeps ONE decoder alive and appends rows to it; a certainty outside [0, 1] fails the
    append BEFORE the graph grows (bp.append_rows validates first) and the accumulator drops its live decoder instead
    of re-entering the failed append on every later decode."""
    driver = importlib.import_module("sca-ldpc_amd.driver")
    rng = np.random.RandomState(5)
    N, W, omega = 499, 7, 5
    first = np.sort(rng.choice(N, W, replace=False))
    ysp = list(rng.choice(N, omega, replace=False))
    base = live()
    with driver.HqcCheckAccumulator(N, first, omega, max_iter=30) as acc:
        for b in range(40):
            acc.add_check(int(rng.randint(N)), 0, 0.9)
        acc.decode(ysp)
        assert acc._bpd is not None and live() != base
        acc.add_check(3, 1, 1.5)  # certainty 1.5 -> prior -0.5
        with pytest.raises(ValueError, match="not a probability"):
            acc.decode(ysp)
        assert acc._bpd is None and live() == base  # dropped, nothing left behind
        acc._cert[acc._R - 1] = 0.8  # the caller repairs its data: the next decode builds a fresh decoder
        acc.decode(ysp)
        assert acc._bpd is not None
    assert live() == base  # context manager closed it
