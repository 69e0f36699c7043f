"""Property-based GPU parity for the q-ary min-sum decoder: random +-1 parity-check matrices,
alphabets Q in {3, 5, 7}, pmfs with zero-probability symbols (+inf LLRs), ragged batches
(covers the unrolled, wave-parallel and lane-per-codeword kernels) -- hard decisions
bit-exact with the oracle."""
import importlib
import os

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from helpers import S

pytestmark = pytest.mark.gpu
qary = importlib.import_module("sca-ldpc_amd.qary")

_examples = [0]


def _progress():
    """A line every 25 examples (long soak runs with SCALDPC_PROPERTY_EXAMPLES in the thousands: a GPU box takes minutes of silence for a hang)."""
    _examples[0] += 1
    if _examples[0] % 25 == 0:
        print(f"[q-ary property examples: {_examples[0]}]", flush=True)


@settings(max_examples=int(os.environ.get("SCALDPC_PROPERTY_EXAMPLES", "30")), deadline=None, derandomize=True,
          suppress_health_check=list(HealthCheck))
@given(R=st.integers(2, 10), N=st.integers(6, 24), dc=st.integers(2, 5), B=st.integers(1, 3),
       batch=st.sampled_from([1, 3, 40, 70, 300]), iters=st.integers(1, 6), seed=st.integers(0, 9999),
       zero_frac=st.sampled_from([0.0, 0.1]), wave=st.sampled_from([-1, 0, 1]), var_small=st.integers(0, 1), llr_tiled=st.integers(0, 1),
       dp=st.integers(0, 1))
def test_random_qary_instances(oracle, R, N, dc, B, batch, iters, seed, zero_frac, wave, var_small, llr_tiled, dp):
    _progress()
    rng = np.random.RandomState(seed)
    Q = 2 * B + 1
    H = np.zeros((R, N), dtype=np.int8)
    for r in range(R):
        cols = rng.choice(N, min(dc, N), replace=False)
        H[r, cols] = rng.choice([-1, 1], size=cols.size)
    g = S.TannerGraph.from_dense(H)
    pmf = rng.dirichlet(np.ones(Q) * 0.8, size=(batch, N)).astype(np.float32)
    if zero_frac:
        z = rng.rand(batch, N, Q) < zero_frac
        z[..., B] = False  # keep the zero symbol possible so that every check admits a configuration
        pmf[z] = 0.0
        pmf /= pmf.sum(axis=2, keepdims=True)
    name = f"DecoderN{N}R{R}V{max(1, int(g.col_degrees().max()))}C{int(g.row_degrees().max())}B{B}"
    dec = qary.decoder_class(name)(H, iters)
    # kernel-form knobs: wave / lane mode, register-resident variable update, tiled conversion, min-plus recursion or unrolled enumeration (Q = 3) -- invisible
    dec.configure(wave=wave, var_small=var_small, llr_tiled=llr_tiled, dp=dp)
    with np.errstate(divide="ignore"):
        try:
            ref = oracle.qary_min_sum_batch(g, Q, pmf, iters, threads=4)
        except RuntimeError as e:  # the oracle refuses (no finite configuration): the device must refuse too
            with pytest.raises(Exception):
                dec.min_sum_batch(pmf)
            return
        got = dec.min_sum_batch(pmf)
    assert np.array_equal(got, ref)


@settings(max_examples=int(os.environ.get("SCALDPC_PROPERTY_EXAMPLES", "20")), deadline=None, derandomize=True,
          suppress_health_check=list(HealthCheck))
@given(R=st.integers(2, 9), NB=st.integers(8, 30), batch=st.sampled_from([1, 3, 40, 70, 130]), iters=st.integers(1, 5),
       seed=st.integers(0, 9999), zero_frac=st.sampled_from([0.0, 0.1, 0.3]), full=st.booleans(),
       form=st.sampled_from(["dp", "dp halves", "dp quarters", "tree", "wave", "lane"]), var_small=st.integers(0, 1))
def test_random_special_instances(oracle, R, NB, batch, iters, seed, zero_frac, full, form, var_small):
    """DecoderSpecial at the Kyber alphabets (B = 2, BSUM = 12) on random H = [H' | I]: rows of six coefficient edges (the
    min-plus recursion `k_q_special_check_dp` / the tree walk) mixed with shorter rows (generic wave kernel), signed
    entries, pmfs with impossible symbols on both alphabets (up to 30 % of them: +inf LLRs, NaN messages after the
    variable update), ragged batches, every check-kernel form -- symbols bit-exact with the oracle
    (decoder_special.rs:471-617 restated)."""
    _progress()
    rng = np.random.RandomState(seed)
    B, SW = 2, 6
    BSUM = SW * B
    Hp = np.zeros((R, NB), dtype=np.int8)
    for r in range(R):
        k = 6 if (full or r % 3) else rng.randint(1, 6)
        cols = rng.choice(NB, k, replace=False)
        Hp[r, cols] = rng.choice([-1, 1], size=k)
    H = np.concatenate([Hp, np.eye(R, dtype=np.int8)], axis=1)
    g = S.TannerGraph.from_dense(H)
    pb = rng.dirichlet(np.ones(5) * 0.7, size=(batch, NB)).astype(np.float32)
    ps = rng.dirichlet(np.ones(2 * BSUM + 1) * 0.7, size=(batch, R)).astype(np.float32)
    if zero_frac:
        zb = rng.rand(batch, NB, 5) < zero_frac
        zb[..., B] = False
        pb[zb] = 0.0
        pb /= pb.sum(axis=2, keepdims=True)
        zs = rng.rand(batch, R, 2 * BSUM + 1) < zero_frac
        zs[..., BSUM] = False
        ps[zs] = 0.0
        ps /= ps.sum(axis=2, keepdims=True)
    dec = qary.decoder_class(f"DecoderN{NB + R}R{R}SW{SW}")(H, iters)
    dec.configure(var_small=var_small, **{"dp": dict(wave=-1, tree=1, dp=1, dp_min=1, dp_split=0, dp_split2=0),
                                          "dp halves": dict(wave=-1, tree=1, dp=1, dp_min=1, dp_split=0, dp_split2=1 << 20),
                                          "dp quarters": dict(wave=-1, tree=1, dp=1, dp_min=1, dp_split=1 << 20), "tree": dict(wave=-1, tree=1, dp=0),
                                          "wave": dict(wave=1, tree=0, dp=0), "lane": dict(wave=0, tree=0, dp=0)}[form])
    with np.errstate(divide="ignore", invalid="ignore"):
        ref = oracle.qary_special_batch(g, B, BSUM, pb, ps, iters, threads=8)
        got = dec.min_sum_batch(pb, ps)
    dec.close()
    assert np.array_equal(got, ref), form
