"""The identity the q-ary check kernels rest on since round 4 (csrc/scaldpc_qary.hip: QEnum, k_q_special_check_tree):

    min over candidates of fl(S_c - a)  ==  fl( (min over candidates of S_c) - a )          in float32, bit for bit,

because x -> fl(x - a) is monotone non-decreasing (exact subtraction is, and so is rounding to nearest).  The reference
(decoder.rs:621-627, decoder_special.rs:531-554) takes the minimum of the candidates; the kernels take the minimum of the sums and
subtract once.  Checked here on the CPU with NumPy float32 -- IEEE arithmetic, the same as the device's v_sub_f32 / v_min_f32 --
on adversarial inputs: neighbouring floats, huge dynamic range, ties, zeros, +inf sums, and the guard for +inf minima.
(The GPU suite holds the kernels themselves to the reference-form oracle: 88 q-ary tests.)"""
import numpy as np


def reference_form(S, a):
    """beta = min_c fl(S_c - a) with f32::min semantics (NaN candidates ignored), +inf when there is no candidate."""
    with np.errstate(invalid="ignore"):
        cand = (S - a).astype(np.float32)
    cand = cand[~np.isnan(cand)]
    return np.float32(np.inf) if cand.size == 0 else cand.min()


def min_marginal_form(S, a):
    fin = S[np.isfinite(S)]
    if fin.size == 0:
        return np.float32(np.inf)  # no assignment with a finite sum: +inf, never inf - inf
    return np.float32(fin.min() - a)


def test_min_of_differences_is_difference_of_min():
    rng = np.random.RandomState(0)
    checked = 0
    for trial in range(4000):
        n = rng.randint(1, 40)
        kind = trial % 5
        if kind == 0:  # sums that differ in the last bits
            base = np.float32(rng.uniform(0.5, 50.0))
            S = np.nextafter(np.full(n, base, dtype=np.float32), np.float32(np.inf) * rng.choice([-1, 1], size=n).astype(np.float32))
            S = np.where(rng.rand(n) < 0.5, S, base).astype(np.float32)
        elif kind == 1:  # huge dynamic range
            S = (10.0 ** rng.uniform(-30, 30, size=n)).astype(np.float32)
        elif kind == 2:  # many ties and zeros
            S = rng.choice(np.array([0.0, 0.5, 1.0, 1.5, 7.25], dtype=np.float32), size=n)
        elif kind == 3:  # some sums are +inf (an assignment through a zero-probability symbol)
            S = rng.uniform(0, 20, size=n).astype(np.float32)
            S[rng.rand(n) < 0.4] = np.inf
        else:
            S = rng.uniform(0, 100, size=n).astype(np.float32)
        # alpha is one of the summands of every S, so 0 <= a <= min S in exact arithmetic; also try values at and just around it
        lo = np.float32(S[np.isfinite(S)].min()) if np.isfinite(S).any() else np.float32(1.0)
        for a in (np.float32(0.0), lo, np.nextafter(lo, np.float32(0.0)), np.float32(lo * rng.rand()), np.float32(lo / 3)):
            want = reference_form(S, np.float32(a))
            got = min_marginal_form(S, np.float32(a))
            assert want.tobytes() == got.tobytes() or (want == 0 and got == 0), (S, a, want, got)
            checked += 1
    assert checked == 20000


def test_infinite_alpha_never_forms_inf_minus_inf():
    """A symbol with alpha = +inf is a summand of every sum through it: all of them are +inf, the minimum stays +inf, and the
    kernels write +inf for it -- the reference's candidates there are inf - inf = NaN, which f32::min ignores: +inf as well."""
    S = np.array([np.inf, np.inf], dtype=np.float32)
    assert reference_form(S, np.float32(np.inf)) == np.inf and min_marginal_form(S, np.float32(np.inf)) == np.inf
