"""The HIP path against exact inference on cycle-free graphs (tests/exact.py) -- no oracle in between.

Same answer key and cases as tests/test_exact_inference.py: BP is exact on trees, so posterior LLRs
of the tanh rule are the true marginals (held to 2e-5 * (1 + |L|): five times inside the product
path's stated fp32 tolerance), min-sum posteriors are the true min-cost differences (fp32 rounding of a
handful of additions), hard decisions are exact, and the q-ary decoders return the enumerated optimum.
Every kernel family is held to the key: LDS-resident single launch, 64-codeword tiles, row-parallel;
for the q-ary decoders the lane-per-codeword, wave-per-check, unrolled and tree-walk forms.
"""
import importlib

import numpy as np
import pytest

import exact
from test_exact_inference import large_qary_tree_case, large_special_tree_case, large_tree_case, qary_tree_case, rep_code_case, special_tree_case

pytestmark = pytest.mark.gpu
S = importlib.import_module("sca-ldpc_amd")
bp = importlib.import_module("sca-ldpc_amd.bp")
qary = importlib.import_module("sca-ldpc_amd.qary")

# tanh rule: the product path's stated fp32 tolerance is 1e-4 * max(1, |L|); against the exact marginals it
# measures 1.5e-6 on these cases (all three kernel families), so it is held to 2e-5 here
TOL = {"product_sum": (2e-5, 2e-5, "sp"), "min_sum": (4e-6, 4e-6, "ms")}


def hip_decode(H, probs, synds, method, iters, path):
    with np.errstate(divide="ignore"):
        dec = bp.bp_decoder(S.TannerGraph.from_dense(H), max_iter=iters, bp_method=method, channel_probs=probs)
    dec.configure(path=path)
    out = dec.decode_batch(synds, early_exit=False, want_llr=True)
    dec.close()
    return out


@pytest.mark.parametrize("path", ["auto", "stream", "edge"])
@pytest.mark.parametrize("method", ["product_sum", "min_sum"])
@pytest.mark.parametrize("hard", [0, 1, 3])
def test_rep_code_every_syndrome(method, hard, path):
    """rep_code(13) -- the reference's doctest code (decode.py:139-149), a path graph -- non-uniform priors
    incl. p = 0 / 1, every syndrome the priors allow, in one batch."""
    H, probs, synds, ex = rep_code_case(hard)
    if path == "edge":  # the row-parallel kernels take up to 64 codewords
        synds, ex = synds[::67][:64], {k: v[::67][:64] for k, v in ex.items()}
    rtol, atol, key = TOL[method]
    got = hip_decode(H, probs, synds, method, 26, path)
    exact.check_binary_llr(got["llr"], got["bits"], ex[key], rtol, atol, f"{method} {path}")
    assert got["converged"].all()


@pytest.mark.parametrize("path", ["auto", "stream", "edge"])
@pytest.mark.parametrize("method", ["product_sum", "min_sum"])
def test_random_trees(method, path):
    worst = 0.0
    for seed in range(12):
        rng = np.random.RandomState(500 + seed)
        n = int(rng.randint(5, 17))
        H = exact.random_binary_tree(rng, n, max_check_degree=int(rng.randint(3, 7)))
        probs = exact.random_priors(rng, n, hard=int(seed % 3 == 2) * 2)
        synds = exact.feasible_syndromes(rng, H, probs, 24)
        ex = exact.binary_exact(H, probs, synds)
        rtol, atol, key = TOL[method]
        got = hip_decode(H, probs, synds, method, 2 * (H.shape[0] + n), path)
        worst = max(worst, exact.check_binary_llr(got["llr"], got["bits"], ex[key], rtol, atol, f"{method} {path} seed {seed}"))
    print(f"{method} {path}: worst |dL| vs exact = {worst:.3e}")


@pytest.mark.parametrize("path", ["auto", "edge"])
@pytest.mark.parametrize("method", ["product_sum", "min_sum"])
def test_large_tree(method, path):
    """A cycle-free graph of 6000 variables (E = 7984: too large for the LDS-resident decoder, so `auto` takes the
    64-codeword-tile kernels the BASELINE configs run on, with their degree buckets, two stream lanes and, here, the first
    iteration without its check pass) against the exact posteriors of tests/exact.tree_exact_binary: 70 codewords on the
    tile kernels, 5 on the row-parallel ones; +-inf priors included.  Fixed-iteration runs are compared on values; early-exit
    runs on their flags (truthful), and on values for the codewords that never satisfied their syndrome and therefore ran
    every iteration."""
    g, probs, synds, ex, _ = large_tree_case(6000, 70, seed=321, hard=6)
    rtol, atol, key = TOL[method]
    nb = 70 if path == "auto" else 5
    with np.errstate(divide="ignore"):
        dec = bp.bp_decoder(g, max_iter=150, bp_method=method, channel_probs=probs)
    dec.configure(path=path)
    got = dec.decode_batch(synds[:nb], early_exit=False, want_llr=True)
    assert dec.last_stats()["row_parallel"] == (nb if path == "edge" else 0)
    worst = exact.check_binary_llr(got["llr"], got["bits"], ex[key][:nb], rtol, atol, f"{method} {path} large tree")
    # (bit-wise MAP decisions need not form a word that satisfies the checks: the flag must simply tell the truth)
    truth = (g.syndrome(got["bits"]) == synds[:nb]).all(axis=1)
    assert np.array_equal(got["converged"].astype(bool), truth)
    early = dec.decode_batch(synds[:nb], early_exit=True, want_llr=True)
    assert np.array_equal(early["converged"].astype(bool), (g.syndrome(early["bits"]) == synds[:nb]).all(axis=1))
    assert (early["iters"] >= 1).all() and (early["iters"] <= 150).all() and (early["iters"][~early["converged"].astype(bool)] == 150).all()
    stuck = ~early["converged"].astype(bool)  # a codeword that never satisfied its syndrome ran all 150 iterations: exact again
    if stuck.any():
        exact.check_binary_llr(early["llr"][stuck], early["bits"][stuck], ex[key][:nb][stuck], rtol, atol, f"{method} {path} early exit")
    dec.close()
    print(f"{method} {path}: 6000-variable tree, worst |dL| vs exact = {worst:.3e}")


def test_single_decode_attributes_are_the_exact_marginals():
    """The ldpc surface itself: `decode()` then `.log_prob_ratios` (float64 [n], log p0 / p1).  With early exit
    the loop stops at the first iteration whose decision satisfies the syndrome, which may come before the
    messages have crossed a deep tree; a star (one check over all variables, depth 1) is exact after the
    first iteration, whenever the loop stops."""
    rng = np.random.RandomState(77)
    n = 9
    H = np.zeros((1, n), dtype=np.int8)
    H[0, :] = 1  # one check over all variables: depth 1
    probs = exact.random_priors(rng, n)
    for s in (0, 1):
        ex = exact.binary_exact(H, probs, np.array([[s]], dtype=np.uint8))
        dec = bp.bp_decoder(H, max_iter=5, bp_method="product_sum", channel_probs=probs, input_vector_type="syndrome")
        out = dec.decode(np.array([s]))
        exact.check_binary_llr(dec.log_prob_ratios[None], out[None], ex["sp"], 1e-4, 1e-4, "decode()")
        dec.close()


# ------------------------------------------------------------------------------- q-ary
@pytest.mark.parametrize("knobs", [dict(), dict(dp=0), dict(wave=0), dict(wave=1), dict(unroll=0, wave=0), dict(unroll=0, wave=1)])
@pytest.mark.parametrize("B", [1, 2])
def test_qary_min_sum_finds_the_minimum_cost_assignment(B, knobs):
    for seed in range(10):
        H, pmf, best = qary_tree_case(seed, B, batch=70 if seed == 0 else 6)
        R, N = H.shape
        nz = H != 0
        name = f"DecoderN{N}R{R}V{int(nz.sum(axis=0).max())}C{int(nz.sum(axis=1).max())}B{B}"
        dec = qary.decoder_class(name)(H, 2 * (R + N))
        dec.configure(**knobs)
        got = dec.min_sum_batch(pmf)
        dec.close()
        assert np.array_equal(got, best), (seed, knobs)


@pytest.mark.parametrize("knobs", [dict(), dict(dp_min=1, dp_split=0, dp_split2=0), dict(dp_min=1, dp_split=0, dp_split2=1 << 20), dict(dp_min=1, dp_split=1 << 20), dict(dp=0), dict(dp=0, tree=0), dict(wave=0), dict(wave=0, tree=0)])
def test_special_decoder_finds_the_minimum_cost_assignment(knobs):
    """DecoderSpecial, B = 2, BSUM = 12 (DecoderN*R*SW6, lib.rs:54-75) on cycle-free [H' | +-I]; rows of 6
    coefficient edges go through the tree-walk kernel, the others through the wave / lane forms."""
    for seed in range(12):
        H, pmf_b, pmf_s, best = special_tree_case(seed, batch=66 if seed == 2 else 4)
        R, N = H.shape
        dec = qary.decoder_class(f"DecoderN{N}R{R}SW6")(H, 2 * (R + N))
        dec.configure(**knobs)
        got = dec.min_sum_batch(pmf_b, pmf_s)
        dec.close()
        assert np.array_equal(got, best), (seed, knobs)


@pytest.mark.parametrize("knobs", [dict(), dict(wave=0), dict(wave=1, unroll=0), dict(var_small=0, llr_tiled=0)])
@pytest.mark.parametrize("B", [1, 2])
def test_qary_large_tree(B, knobs):
    """A +-1 tree of 250 variables (checks of up to 5 edges): the decoder's symbols are the arg-minima of the exact
    min-marginals (tests/exact.tree_exact_qary: (min, +) elimination over integer partial sums, float64) wherever those
    are clear of fp32 rounding -- every kernel form, 70 channel outputs."""
    H, pmf, dec_exact, ok = large_qary_tree_case(250, B, 70, seed=40 + B)
    R, N = H.shape
    nz = H != 0
    d = qary.decoder_class(f"DecoderN{N}R{R}V{int(nz.sum(axis=0).max())}C{int(nz.sum(axis=1).max())}B{B}")(H, 80)
    d.configure(**knobs)
    got = d.min_sum_batch(pmf)
    d.close()
    assert ok.mean() > 0.9 and np.array_equal(got[ok], dec_exact[ok])


@pytest.mark.parametrize("knobs", [dict(), dict(dp_min=1, dp_split=0, dp_split2=0), dict(dp_min=1, dp_split=0, dp_split2=1 << 20), dict(dp_min=1, dp_split=1 << 20), dict(dp=0), dict(dp=0, tree=0), dict(wave=0), dict(var_small=0, llr_tiled=0)])
def test_special_large_tree(knobs):
    """DecoderSpecial (B = 2, BSUM = 12) on a cycle-free [H' | +-I] of 40 checks with 2-6 coefficient edges each: symbols =
    arg-minima of the exact min-marginals wherever those are clear; tree-walk, wave and lane kernels; 66 channel outputs."""
    H, pb, ps, dec_exact, ok = large_special_tree_case(40, 66, seed=70)
    R, N = H.shape
    d = qary.decoder_class(f"DecoderN{N}R{R}SW6")(H, 100)
    d.configure(**knobs)
    got = d.min_sum_batch(pb, ps)
    d.close()
    assert ok.mean() > 0.85 and np.array_equal(got[ok], dec_exact[ok])
