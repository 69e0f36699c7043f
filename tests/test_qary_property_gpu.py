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


@settings(max_examples=int(os.environ.get("SCALDPC_PROPERTY_EXAMPLES", "30")), deadline=None, derandomize=True,
          suppress_health_check=list(HealthCheck))
@given(R=st.integers(2, 10), N=st.integers(6, 24), dc=st.integers(2, 5), B=st.integers(1, 3),
       batch=st.sampled_from([1, 3, 40, 70, 300]), iters=st.integers(1, 6), seed=st.integers(0, 9999),
       zero_frac=st.sampled_from([0.0, 0.1]), wave=st.sampled_from([-1, 0, 1]), var_small=st.integers(0, 1), llr_tiled=st.integers(0, 1))
def test_random_qary_instances(oracle, R, N, dc, B, batch, iters, seed, zero_frac, wave, var_small, llr_tiled):
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
    dec.configure(wave=wave, var_small=var_small, llr_tiled=llr_tiled)  # kernel-form knobs: wave / lane mode, register-resident variable update, tiled conversion -- invisible
    with np.errstate(divide="ignore"):
        try:
            ref = oracle.qary_min_sum_batch(g, Q, pmf, iters, threads=4)
        except RuntimeError as e:  # the oracle refuses (no finite configuration): the device must refuse too
            with pytest.raises(Exception):
                dec.min_sum_batch(pmf)
            return
        got = dec.min_sum_batch(pmf)
    assert np.array_equal(got, ref)
