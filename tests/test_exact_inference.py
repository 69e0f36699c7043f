"""The CPU oracle against exact inference on cycle-free graphs (tests/exact.py).

This pins what no fixture of the reference pins: posterior LLR VALUES of the binary decoder (both
rules), the min-sum rule as a whole (the reference never selects it), and DecoderSpecial (its own
tests are commented out, decoder_special.rs:691-746).  BP is exact on trees, so the answer key is
plain enumeration and shares nothing with the oracle's sweeps.  tests/test_exact_inference_gpu.py
holds the HIP path to the same key.
"""
import importlib

import numpy as np
import pytest

import exact

S = importlib.import_module("sca-ldpc_amd")

# oracle instantiations and the tolerance each is held to: |dL| <= atol + rtol * |L|.
#   float64: rounding only.  float32 min-sum: adds of fp32 prior LLRs (log((1-p)/p) rounded to fp32, then a few
#   additions).  float32 tanh rule: the stated fp32 tolerance of the product path (1e-4 * max(1, |L|)).
#   The ratio-domain and textbook forms lose digits to the cancellation in 2 / (1 + r) - 1 and 1 - tanh as messages
#   grow (measured 1.2e-7 at |L| = 11 in float64) -- the reason the build's kernels use the complement form.
SUM_PRODUCT = [("product_sum", "f64", 1e-6, 1e-6), ("product_sum_log", "f64", 1e-6, 1e-6), ("tanh_complement", "f64", 1e-9, 1e-9),
               ("tanh_complement", "f32", 1e-4, 1e-4)]
MIN_SUM = [("min_sum", "f64", 1e-10, 1e-10), ("min_sum", "f32", 4e-6, 4e-6)]


def run_oracle(oracle, H, probs, synds, method, dtype, iters):
    g = S.TannerGraph.from_dense(H)
    with np.errstate(divide="ignore", invalid="ignore"):
        return oracle.bp_decode_batch(g, probs, synds, 0, iters, method, dtype=dtype, threads=4, early_exit=False)


def rep_code_case(hard):
    """rep_code(13), the reference's own doctest code (decode.py:139-149) -- a path graph -- with
    non-uniform priors on both sides of 1/2, `hard` of them exactly 0 / 1, and EVERY syndrome the priors allow."""
    rng = np.random.RandomState(13 + hard)
    H = S.codes.rep_code_graph(13).to_dense(np.int8)
    probs = exact.random_priors(rng, 13, hard=hard)
    synds = exact.all_binary_vectors(12)
    ex = exact.binary_exact(H, probs, synds)
    keep = ex["feasible"]
    assert keep.sum() == (4096 >> max(hard - 1, 0))  # k pinned bits leave 2^(12-(k-1)) syndromes possible
    return H, probs, synds[keep], {k: v[keep] for k, v in ex.items()}


@pytest.mark.parametrize("hard", [0, 1, 3])
def test_rep_code_every_syndrome(oracle, hard):
    H, probs, synds, ex = rep_code_case(hard)
    for method, dtype, rtol, atol in SUM_PRODUCT:
        r = run_oracle(oracle, H, probs, synds, method, dtype, 26)
        exact.check_binary_llr(r["llr"], r["bits"], ex["sp"], rtol, atol, f"{method}/{dtype}")
        assert r["converged"].all()
    for method, dtype, rtol, atol in MIN_SUM:
        r = run_oracle(oracle, H, probs, synds, method, dtype, 26)
        exact.check_binary_llr(r["llr"], r["bits"], ex["ms"], rtol, atol, f"{method}/{dtype}")


@pytest.mark.parametrize("seed", range(12))
def test_random_trees(oracle, seed):
    rng = np.random.RandomState(500 + seed)
    n = int(rng.randint(5, 17))
    H = exact.random_binary_tree(rng, n, max_check_degree=int(rng.randint(3, 7)))
    assert exact.is_forest(H) and (H.sum(axis=1) >= 2).all()
    probs = exact.random_priors(rng, n, hard=int(seed % 3 == 2) * 2)
    synds = exact.feasible_syndromes(rng, H, probs, 24)
    ex = exact.binary_exact(H, probs, synds)
    assert ex["feasible"].all()
    iters = 2 * (H.shape[0] + n)
    for method, dtype, rtol, atol in SUM_PRODUCT:
        r = run_oracle(oracle, H, probs, synds, method, dtype, iters)
        exact.check_binary_llr(r["llr"], r["bits"], ex["sp"], rtol, atol, f"{method}/{dtype} seed {seed}")
    for method, dtype, rtol, atol in MIN_SUM:
        r = run_oracle(oracle, H, probs, synds, method, dtype, iters)
        exact.check_binary_llr(r["llr"], r["bits"], ex["ms"], rtol, atol, f"{method}/{dtype} seed {seed}")


def test_the_answer_key_itself():
    """Two variables, one check: the marginals can be written down by hand."""
    H = np.array([[1, 1]], dtype=np.int8)
    p = np.array([0.1, 0.3])
    ex = exact.binary_exact(H, p, np.array([[0], [1]], dtype=np.uint8))
    # s = 0: e in {00, 11}: P = .9*.7, .1*.3 ; s = 1: e in {10, 01}: P = .1*.7, .9*.3
    assert np.allclose(ex["sp"][0], np.log(0.63 / 0.03)) and np.allclose(ex["ms"][0], np.log(0.63 / 0.03))
    assert np.allclose(ex["sp"][1], [np.log(0.27 / 0.07), np.log(0.07 / 0.27)])
    assert not exact.is_forest(np.array([[1, 1], [1, 1]])) and exact.is_forest(np.array([[1, 1, 0], [0, 1, 1]]))
    Hq = np.array([[1, -1, 0], [0, 1, 1]], dtype=np.int8)  # x0 = x1, x2 = -x1
    llr = [np.array([2.0, 0.5, 0.0]), np.array([0.0, 1.0, 3.0]), np.array([0.0, 0.2, 4.0])]
    best, cost, gap = exact.qary_exact(Hq, llr, [1, 1, 1])
    # candidates (x1 = -1, 0, 1): (-1,-1,1): 2+0+4 = 6; (0,0,0): .5+1+.2 = 1.7; (1,1,-1): 0+3+0 = 3
    assert list(best) == [0, 0, 0] and np.isclose(cost, 1.7) and np.isclose(gap, 1.3)


# ------------------------------------------------------------------------------- q-ary
def qary_tree_case(seed, B, batch=6):
    """A cycle-free H with +-1 entries and `batch` channel outputs whose optimum is unique by a clear margin."""
    rng = np.random.RandomState(900 + 10 * B + seed)
    Q = 2 * B + 1
    n = int(rng.randint(4, 10 if B == 1 else 7))
    H = exact.random_qary_tree(rng, n, max_check_degree=4)
    pmfs, bests = [], []
    while len(pmfs) < batch:
        pmf = rng.dirichlet(np.ones(Q) * 1.5, size=n).astype(np.float32)
        if (seed + len(pmfs)) % 4 == 3:  # zero-probability symbols: +inf costs (decoder.rs:688)
            pmf[::3, 0] = 0.0
            pmf = (pmf / pmf.sum(axis=1, keepdims=True)).astype(np.float32)
        best, cost, gap = exact.qary_exact(H, exact.pmf_to_llr64(pmf), [B] * n)
        if gap > 1e-3:  # a unique optimum, clear of fp32 rounding
            pmfs.append(pmf)
            bests.append(best)
    return H, np.stack(pmfs), np.stack(bests)


@pytest.mark.parametrize("B", [1, 2])
@pytest.mark.parametrize("seed", range(10))
def test_qary_min_sum_finds_the_minimum_cost_assignment(oracle, B, seed):
    H, pmf, best = qary_tree_case(seed, B)
    assert exact.is_forest(H)
    g = S.TannerGraph.from_dense(H)
    with np.errstate(divide="ignore"):
        got = oracle.qary_min_sum_batch(g, 2 * B + 1, pmf, 2 * sum(H.shape))
    assert np.array_equal(got, best), (got, best)


SPECIAL_SHAPES = [(2, [3, 3]), (3, [3, 2, 3]), (2, [6, 2]), (3, [2, 4, 2]), (1, [6]), (2, [4, 4])]


def special_tree_case(seed, batch=4):
    rng = np.random.RandomState(1300 + seed)
    B, SW = 2, 6
    R, coeffs = SPECIAL_SHAPES[seed % len(SPECIAL_SHAPES)]
    H = exact.random_special_tree(rng, R, coeffs)
    if seed % 2:
        H[:, H.shape[1] - R:] *= rng.choice(np.array([-1, 1], dtype=np.int8), size=R)[None, :]  # -1 on the identity part too
    BV = H.shape[1] - R
    pb, ps, bests = [], [], []
    while len(pb) < batch:
        pmf_b = rng.dirichlet(np.ones(2 * B + 1) * 1.5, size=BV).astype(np.float32)
        pmf_s = rng.dirichlet(np.ones(2 * SW * B + 1) * 0.8, size=R).astype(np.float32)
        best, cost, gap = exact.qary_exact_special(H, exact.pmf_to_llr64(pmf_b), exact.pmf_to_llr64(pmf_s), B, SW * B)
        if gap > 1e-3:
            pb.append(pmf_b)
            ps.append(pmf_s)
            bests.append(best)
    return H, np.stack(pb), np.stack(ps), np.stack(bests)


@pytest.mark.parametrize("seed", range(12))
def test_special_decoder_finds_the_minimum_cost_assignment(oracle, seed):
    """DecoderSpecial (decoder_special.rs:471-617), B = 2, BSUM = 12 (the SW6 Kyber classes of lib.rs:54-75), on
    cycle-free [H' | +-I] with 1-3 checks of 2-6 coefficient edges: the decision is the enumerated optimum."""
    H, pmf_b, pmf_s, best = special_tree_case(seed)
    assert exact.is_forest(H)
    g = S.TannerGraph.from_dense(H)
    got = oracle.qary_special_batch(g, 2, 12, pmf_b, pmf_s, 2 * sum(H.shape))
    assert np.array_equal(got, best), (got, best)


# ------------------------------------------------------------------------ large trees (no enumeration)
def large_tree_case(n_vars, batch, seed, hard=0):
    """A random cycle-free graph with thousands of variables, priors on both sides of 1/2, syndromes the priors allow, and
    the exact posteriors from tests/exact.tree_exact_binary (elimination in the log-probability domain, float64).
    Returns (TannerGraph, dense-free rows, probs, synds, exact dict, iterations that certainly suffice)."""
    rng = np.random.RandomState(seed)
    rows, n = exact.random_binary_tree_sparse(rng, n_vars, max_check_degree=6)
    g = S.TannerGraph.from_row_supports(rows, n)
    H = g.to_dense(np.int8)
    probs = exact.random_priors(rng, n, hard=hard)
    e = (rng.rand(batch, n) < 0.5).astype(np.uint8)
    e[:, probs == 0.0] = 0
    e[:, probs == 1.0] = 1
    synds = g.syndrome(e)
    ex = exact.tree_exact_binary(H, probs, synds)
    return g, probs, synds, ex, 2 * (g.m + 2)  # (flooding needs at most the diameter; the node count bounds it)


def test_the_two_answer_keys_agree():
    """tree_exact_binary (elimination, any size) against binary_exact (enumeration) where both apply."""
    rng = np.random.RandomState(5)
    for trial in range(8):
        n = int(rng.randint(5, 15))
        H = exact.random_binary_tree(rng, n, 5)
        probs = exact.random_priors(rng, n, hard=(trial % 3 == 2) * 2)
        synds = exact.feasible_syndromes(rng, H, probs, 7)
        a, b = exact.binary_exact(H, probs, synds), exact.tree_exact_binary(H, probs, synds)
        for k in ("sp", "ms"):
            fin = np.isfinite(a[k])
            assert np.array_equal(np.isfinite(b[k]), fin) and np.array_equal(np.sign(a[k][~fin]), np.sign(b[k][~fin]))
            assert np.allclose(a[k][fin], b[k][fin], rtol=1e-10, atol=1e-10)


@pytest.mark.parametrize("hard", [0, 5])
def test_large_tree(oracle, hard):
    """700 variables, depth far beyond what enumeration reaches: every oracle method against the exact posteriors."""
    g, probs, synds, ex, _ = large_tree_case(700, 9, seed=77 + hard, hard=hard)
    # iterations: the tree's depth is O(log n) for this construction; 120 is far more than enough and keeps the test quick
    # measured: float64 methods 1e-13 relative, float32 5e-7 (also at 6000 variables); held to 1e-9 / 2e-5
    for methods, key in ((SUM_PRODUCT, "sp"), (MIN_SUM, "ms")):
        for method, dtype, _, _ in methods:
            tol = 1e-9 if dtype == "f64" else 2e-5
            with np.errstate(divide="ignore", invalid="ignore"):
                r = oracle.bp_decode_batch(g, probs, synds, 0, 120, method, dtype=dtype, threads=4, early_exit=False)
            exact.check_binary_llr(r["llr"], r["bits"], ex[key], tol, tol, f"{method}/{dtype} large tree")


def large_qary_tree_case(n_vars, B, batch, seed):
    """A +-1 tree of a few hundred variables with the exact min-marginals of every variable (tests/exact.tree_exact_qary).
    Returns (H int8, pmf float32 [batch, N, Q], decisions int8 [batch, N], decided bool [batch, N]): `decided` marks the
    variables whose best symbol beats the second best by a margin clear of fp32 rounding (sums of N costs)."""
    rng = np.random.RandomState(seed)
    Q = 2 * B + 1
    rows, n = exact.random_binary_tree_sparse(rng, n_vars, max_check_degree=5)
    H = np.zeros((len(rows), n), dtype=np.int8)
    for r, cs in enumerate(rows):
        H[r, cs] = rng.choice(np.array([-1, 1], dtype=np.int8), size=len(cs))
    pmf = rng.dirichlet(np.ones(Q) * 1.2, size=(batch, n)).astype(np.float32)
    dec = np.zeros((batch, n), dtype=np.int8)
    ok = np.zeros((batch, n), dtype=bool)
    for b in range(batch):
        mm = exact.tree_exact_qary(H, exact.pmf_to_llr64(pmf[b]), B)
        srt = np.sort(mm, axis=1)
        dec[b] = mm.argmin(axis=1) - B
        ok[b] = srt[:, 1] - srt[:, 0] > 1e-3 * max(1.0, float(srt[:, 0].max()))
    return H, pmf, dec, ok


def test_the_two_qary_answer_keys_agree():
    rng = np.random.RandomState(9)
    for trial in range(8):
        B = 1 + trial % 2
        n = int(rng.randint(4, 9 if B == 1 else 7))
        H = exact.random_qary_tree(rng, n, 4)
        llr = exact.pmf_to_llr64(rng.dirichlet(np.ones(2 * B + 1) * 1.5, size=n))
        best, cost, gap = exact.qary_exact(H, llr, [B] * n)
        mm = exact.tree_exact_qary(H, llr, B)
        assert np.allclose(mm.min(axis=1), cost) and (gap < 1e-9 or np.array_equal(mm.argmin(axis=1) - B, best))


@pytest.mark.parametrize("B", [1, 2])
def test_qary_large_tree(oracle, B):
    """250 variables: the oracle's symbols are the arg-minima of the exact min-marginals wherever those are clear."""
    H, pmf, dec, ok = large_qary_tree_case(250, B, 6, seed=40 + B)
    g = S.TannerGraph.from_dense(H)
    got = oracle.qary_min_sum_batch(g, 2 * B + 1, pmf, 80, threads=4)
    assert ok.mean() > 0.9 and np.array_equal(got[ok], dec[ok])


def large_special_tree_case(R, batch, seed):
    """DecoderSpecial on a cycle-free [H' | +-I] of R checks (2-6 coefficient edges each, chained through shared
    coefficient variables), B = 2, BSUM = 12, with the exact min-marginals (per-variable alphabets)."""
    rng = np.random.RandomState(seed)
    coeffs = [int(rng.randint(2, 7)) for _ in range(R)]
    coeffs[0] = 6  # (at least one row of six coefficient edges: the tree-walk kernel's shape)
    H = exact.random_special_tree(rng, R, coeffs)
    H[:, H.shape[1] - R:] *= rng.choice(np.array([-1, 1], dtype=np.int8), size=R)[None, :]
    BV = H.shape[1] - R
    pb = rng.dirichlet(np.ones(5) * 1.2, size=(batch, BV)).astype(np.float32)
    ps = rng.dirichlet(np.ones(25) * 0.6, size=(batch, R)).astype(np.float32)
    dec = np.zeros((batch, H.shape[1]), dtype=np.int8)
    ok = np.zeros((batch, H.shape[1]), dtype=bool)
    alph = [2] * BV + [12] * R
    for b in range(batch):
        llr = [exact.pmf_to_llr64(pb[b][v]) for v in range(BV)] + [exact.pmf_to_llr64(ps[b][r]) for r in range(R)]
        mm = exact.tree_exact_qary(H, llr, alph)
        top = max(float(np.min(m)) for m in mm)
        for v, m in enumerate(mm):
            srt = np.sort(m)
            dec[b, v] = int(np.argmin(m)) - alph[v]
            ok[b, v] = srt[1] - srt[0] > 1e-3 * max(1.0, top)
    return H, pb, ps, dec, ok


def test_special_large_tree(oracle):
    """40 checks, ~120 coefficient variables: far beyond enumeration (5^120); the oracle's symbols are the arg-minima of the
    exact min-marginals wherever those are clear."""
    H, pb, ps, dec, ok = large_special_tree_case(40, 5, seed=70)
    assert exact.is_forest(H)
    g = S.TannerGraph.from_dense(H)
    got = oracle.qary_special_batch(g, 2, 12, pb, ps, 100, threads=4)
    assert ok.mean() > 0.85 and np.array_equal(got[ok], dec[ok])
