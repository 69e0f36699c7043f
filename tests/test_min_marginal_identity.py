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


# ---------------------------------------------------------------------------------------------------------------------
# The min-plus recursion of k_q_special_check_dp (csrc/scaldpc_qary_special.h): no enumeration at all.
#     S = fl(..fl(fl(0 + a_0[d_0]) + a_1[d_1]).. + a_s[-sum d])       is a chain of monotone steps x -> fl(x + c),
# so among the assignments that agree from some edge on and have the same digit sum before it, the one with the smallest
# partial sum has the smallest S: the minimum of S over a class of prefixes is S continued from the class's minimal
# partial sum.  Checked here against the reference-form enumeration (decoder_special.rs:531-554) in NumPy float32, bit for
# bit, on small checks (2 - 4 coefficient edges, alphabets of 3 and 5) with the same adversarial values.
# ---------------------------------------------------------------------------------------------------------------------
def _fmin_ignoring_nan(x, axis=None):
    """f32::min over an axis: NaN candidates ignored, +inf where there is none."""
    y = np.where(np.isnan(x), np.float32(np.inf), x)
    return y.min(axis=axis) if axis is not None else y.min()


def special_check_enumerated(a, a_sum, B, BSUM):
    """Reference form: every assignment forms S left to right; beta_j[d] = min of fl(S - a_j[d]); beta_s likewise."""
    nb, QB = a.shape
    with np.errstate(invalid="ignore", over="ignore"):
        S = np.zeros((1,) * nb, dtype=np.float32)
        dsum = np.zeros((1,) * nb, dtype=np.int64)
        for j in range(nb):
            shape = [1] * nb
            shape[j] = QB
            S = (S + a[j].reshape(shape)).astype(np.float32)
            dsum = dsum + (np.arange(QB) - B).reshape(shape)
        t = -dsum + BSUM
        S = (S + a_sum[t]).astype(np.float32)
        beta = np.empty_like(a)
        for j in range(nb):
            shape = [1] * nb
            shape[j] = QB
            cand = (S - a[j].reshape(shape)).astype(np.float32)
            beta[j] = _fmin_ignoring_nan(np.moveaxis(cand, j, 0).reshape(QB, -1), axis=1)
        beta_s = np.full(a_sum.shape, np.inf, dtype=np.float32)
        cand = (S - a_sum[t]).astype(np.float32)
        for tt in np.unique(t):
            beta_s[tt] = _fmin_ignoring_nan(cand[t == tt])
    return beta, beta_s


def _minplus_step(P, ak):
    """out[u] = min over q of fl(P[u - q] + ak[q]), NaN candidates ignored (NaN where there is nothing else, like v_min3)."""
    QB = ak.size
    out = np.full(P.size + QB - 1, np.nan, dtype=np.float32)
    with np.errstate(invalid="ignore", over="ignore"):
        for u in range(out.size):
            c = np.array([np.float32(P[u - q] + ak[q]) for q in range(QB) if 0 <= u - q < P.size], dtype=np.float32)
            c = c[~np.isnan(c)]
            if c.size:
                out[u] = c.min()
    return out


def special_check_minplus(a, a_sum, B, BSUM):
    nb, QB = a.shape
    top = BSUM + nb * B
    WN = (QB - 1) * nb + 1
    asw = np.array([a_sum[top - w] for w in range(WN)], dtype=np.float32)
    beta = np.empty_like(a)
    beta_s = np.full(a_sum.shape, np.inf, dtype=np.float32)
    P = np.zeros(1, dtype=np.float32)
    with np.errstate(invalid="ignore", over="ignore"):
        for j in range(nb):
            for d in range(QB):
                V = (P + a[j, d]).astype(np.float32)
                for k in range(j + 1, nb):
                    V = _minplus_step(V, a[k])
                c = (V + asw[d : d + V.size]).astype(np.float32)
                c = c[~np.isnan(c)]
                M = c.min() if c.size else np.float32(np.nan)
                beta[j, d] = np.float32(M - a[j, d]) if np.isfinite(M) else np.float32(np.inf)
            P = _minplus_step(P, a[j])
        for w in range(WN):
            M = np.float32(P[w] + asw[w])
            beta_s[top - w] = np.float32(M - asw[w]) if np.isfinite(M) else np.float32(np.inf)
    return beta, beta_s


def _draw(rng, shape, kind):
    if kind == 0:  # every addition rounds: 20 binades
        return (-np.log(rng.rand(*shape) + 1e-7) * 2.0 ** rng.randint(-10, 10, size=shape)).astype(np.float32)
    if kind == 1:  # ties everywhere
        return (0.25 * rng.randint(0, 8, size=shape)).astype(np.float32)
    if kind == 2:  # impossible symbols
        x = (-np.log(rng.rand(*shape) + 1e-7)).astype(np.float32)
        x[rng.rand(*shape) < 0.25] = np.inf
        return x
    if kind == 3:  # NaN alphas (inf - inf of the variable update) and +inf
        x = (3.0 * rng.rand(*shape)).astype(np.float32)
        x[rng.rand(*shape) < 0.15] = np.nan
        x[rng.rand(*shape) < 0.15] = np.inf
        return x
    if kind == 4:  # zeros
        x = rng.rand(*shape).astype(np.float32)
        x[rng.rand(*shape) < 0.3] = 0.0
        return x
    x = (1e30 * rng.rand(*shape)).astype(np.float32)  # sums overflow
    big = rng.rand(*shape) < 0.2
    x[big] = (np.finfo(np.float32).max * (0.3 + 0.5 * rng.rand(*shape))).astype(np.float32)[big]
    return x


def test_minplus_recursion_equals_the_enumeration_bit_for_bit():
    rng = np.random.RandomState(5)
    compared = 0
    for trial in range(360):
        kind = trial % 6
        B = 1 + (trial // 6) % 2
        QB = 2 * B + 1
        nb = 2 + (trial // 12) % 3
        BSUM = nb * B + rng.randint(0, 3)  # (the window of reachable row-sum symbols may be narrower than the alphabet)
        a = _draw(rng, (nb, QB), kind)
        a_sum = _draw(rng, (2 * BSUM + 1,), kind)
        want, want_s = special_check_enumerated(a, a_sum, B, BSUM)
        got, got_s = special_check_minplus(a, a_sum, B, BSUM)
        assert want.tobytes() == got.tobytes(), (kind, a, a_sum, want, got)
        assert want_s.tobytes() == got_s.tobytes(), (kind, a, a_sum, want_s, got_s)
        compared += want.size + want_s.size
    assert compared > 7000


# ---------------------------------------------------------------------------------------------------------------------
# The clipped recursion of k_q_check_dp (csrc/scaldpc_qary_rows.h) for Decoder's constraint (decoder.rs:585-631): the last
# edge's symbol follows from sum d = 0, so an assignment exists only for free-digit sums U in [(K - 2) B, K B], and every table
# keeps only the digit sums that can still reach that window.  The same windows (DpRange) restated here, against the
# reference-form enumeration over the finite supports, bit for bit, rows of 2 .. 6 edges, alphabets of 3 and 5.
# ---------------------------------------------------------------------------------------------------------------------
def generic_check_enumerated(a, B):
    """decoder.rs:585-631: finite symbols of the first K - 1 edges enumerated, the last follows; S left to right; an assignment
    counts if its S is finite.  Returns (beta [K][Q], has_configuration)."""
    import itertools

    K, Q = a.shape
    beta = np.full((K, Q), np.inf, dtype=np.float32)
    fin = [[q for q in range(Q) if np.isfinite(a[j, q])] for j in range(K)]
    nconf = 0
    with np.errstate(invalid="ignore", over="ignore"):
        for qs in itertools.product(*fin[: K - 1]):
            dl = -sum(q - B for q in qs)
            if not -B <= dl <= B:
                continue
            S = np.float32(0.0)
            for j, q in enumerate(qs):
                S = np.float32(S + a[j, q])
            S = np.float32(S + a[K - 1, dl + B])
            if not np.isfinite(S):
                continue
            nconf += 1
            for j, q in enumerate(qs + (dl + B,)):
                beta[j, q] = min(beta[j, q], np.float32(S - a[j, q]))
    return beta, nconf > 0


def generic_check_minplus_clipped(a, B):
    K, Q = a.shape
    NB, S_, TL, TH = K - 1, Q - 1, (K - 2) * B, K * B
    plo = lambda k: max(0, TL - (NB - k) * S_)  # noqa: E731  (prefix over k edges)
    phi = lambda k: min(k * S_, TH)  # noqa: E731
    vlo = lambda k, d: max(0, TL - d - (NB - k) * S_)  # noqa: E731  (pinned to d, k edges done)
    vhi = lambda k, d: min((k - 1) * S_, TH - d)  # noqa: E731

    def step(tab, lo_in, ak, lo_out, hi_out):
        out = np.full(hi_out - lo_out + 1, np.inf, dtype=np.float32)
        with np.errstate(invalid="ignore", over="ignore"):
            for u in range(lo_out, hi_out + 1):
                c = [np.float32(tab[u - q - lo_in] + ak[q]) for q in range(Q) if 0 <= u - q - lo_in < tab.size]
                c = [x for x in c if not np.isnan(x)]
                out[u - lo_out] = min(c) if c else np.float32(np.nan)
        return out

    beta = np.full((K, Q), np.inf, dtype=np.float32)
    P = np.zeros(1, dtype=np.float32)  # prefix over 0 edges: digit sum 0
    with np.errstate(invalid="ignore", over="ignore"):
        for j in range(NB):
            for d in range(Q):
                lo, hi = vlo(j + 1, d), vhi(j + 1, d)
                M = np.float32(np.inf)
                if hi >= lo:
                    V = np.array([np.float32(P[u - plo(j)] + a[j, d]) for u in range(lo, hi + 1)], dtype=np.float32)
                    dead = False
                    for k in range(j + 1, NB):
                        lo2, hi2 = vlo(k + 1, d), vhi(k + 1, d)
                        if hi2 < lo2:
                            dead = True
                            break
                        V, lo, hi = step(V, lo, a[k], lo2, hi2), lo2, hi2
                    if not dead:
                        c = [np.float32(V[u - lo] + a[K - 1, K * B - u - d]) for u in range(lo, hi + 1)]
                        c = [x for x in c if not np.isnan(x)]
                        M = min(c) if c else np.float32(np.nan)
                beta[j, d] = np.float32(M - a[j, d]) if np.isfinite(M) else np.float32(np.inf)
            P = step(P, plo(j), a[j], plo(j + 1), phi(j + 1))
        any_conf = False
        for ql in range(Q):
            U = K * B - ql
            M = np.float32(P[U - plo(NB)] + a[K - 1, ql]) if plo(NB) <= U <= phi(NB) else np.float32(np.inf)
            any_conf |= bool(np.isfinite(M))
            beta[K - 1, ql] = np.float32(M - a[K - 1, ql]) if np.isfinite(M) else np.float32(np.inf)
    return beta, any_conf


def test_clipped_minplus_recursion_equals_the_enumeration_bit_for_bit():
    rng = np.random.RandomState(9)
    compared = 0
    for trial in range(420):
        kind = trial % 6
        B = 1 + (trial // 6) % 2
        K = 2 + (trial // 12) % 5
        a = _draw(rng, (K, 2 * B + 1), kind)
        if any(not np.isfinite(a[j]).any() for j in range(K)):
            continue  # (an edge without a finite symbol: the reference would not terminate, the kernels report it)
        want, conf = generic_check_enumerated(a, B)
        got, conf2 = generic_check_minplus_clipped(a, B)
        assert conf == conf2, (kind, a)
        assert want.tobytes() == got.tobytes(), (kind, K, B, a, want, got)
        compared += want.size
    assert compared > 4000
