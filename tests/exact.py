"""Exact inference by enumeration -- an answer key that owes nothing to any decoder.

Belief propagation is EXACT on a cycle-free Tanner graph: after as many flooding iterations as
the graph is deep, sum-product posteriors are the true posterior marginals, min-sum posteriors
are the true min-cost (max-marginal) differences, and the q-ary min-sum decision is the
minimum-cost valid assignment.  The reference holds no fixture for posterior LLR values, for
min-sum (it never selects it) or for DecoderSpecial (its tests are commented out,
decoder_special.rs:691-746); these enumerators pin all three -- the oracle in the CPU suite, the
HIP path in the GPU suite -- on graphs small enough to enumerate (float64, pure NumPy).

Nothing here looks at messages, schedules or kernels: the definitions are
  sum-product  L_i = log  sum_{e: He=s, e_i=0} P(e)  -  log  sum_{e: He=s, e_i=1} P(e)
  min-sum      L_i = min_{e: He=s, e_i=1} cost(e)   -  min_{e: He=s, e_i=0} cost(e),  cost = -log P
  q-ary        x*  = argmin_{x: sum_j H[c,j] x_j = 0 over the INTEGERS for every check c} sum_v llr_v[x_v]
with P(e) = prod_j p_j^e_j (1-p_j)^(1-e_j) (the per-bit priors of ldpc.bp_decoder's channel_probs).
"""

import numpy as np


# --------------------------------------------------------------------------- binary
def all_binary_vectors(n):
    return ((np.arange(1 << n)[:, None] >> np.arange(n)[None, :]) & 1).astype(np.uint8)


def binary_exact(H, probs, synds):
    """H dense [m, n] 0/1 (n <= 18), probs [n] in [0, 1], synds uint8 [batch, m].
    Returns dict(sp [batch, n], ms [batch, n], feasible [batch]): sum-product / min-sum posterior
    LLRs as defined above (+-inf where one side has no feasible vector; rows of infeasible
    syndromes are NaN and flagged)."""
    H = np.asarray(H) & 1
    m, n = H.shape
    assert n <= 18
    E = all_binary_vectors(n)
    key_of = (1 << np.arange(m, dtype=np.int64))
    keys = ((E.astype(np.int64) @ H.T.astype(np.int64)) & 1) @ key_of
    p = np.asarray(probs, dtype=np.float64)
    with np.errstate(divide="ignore"):
        l1, l0 = np.log(p), np.log1p(-p)
    # log P(e): a term with probability 0 makes the vector impossible (-inf); 0 * -inf never formed
    logw = np.where(E == 1, l1[None, :], l0[None, :]).sum(axis=1)
    synds = np.atleast_2d(np.asarray(synds, dtype=np.int64) & 1)
    out_sp = np.full((synds.shape[0], n), np.nan)
    out_ms = np.full((synds.shape[0], n), np.nan)
    feas = np.zeros(synds.shape[0], dtype=bool)
    for b, s in enumerate(synds):
        sel = keys == int(s @ key_of)
        w, e = logw[sel], E[sel]
        if not np.isfinite(w).any():
            continue
        feas[b] = True
        for i in range(n):
            w0, w1 = w[e[:, i] == 0], w[e[:, i] == 1]
            with np.errstate(invalid="ignore"):
                out_sp[b, i] = _lse(w0) - _lse(w1)
                out_ms[b, i] = _max(w0) - _max(w1)  # = min cost(e_i=1) - min cost(e_i=0)
    return {"sp": out_sp, "ms": out_ms, "feasible": feas}


def _lse(w):
    w = w[np.isfinite(w)]
    if w.size == 0:
        return -np.inf
    mx = w.max()
    return mx + np.log(np.exp(w - mx).sum())


def _max(w):
    return w.max() if w.size else -np.inf


def random_binary_tree(rng, n_vars, max_check_degree=4):
    """Random cycle-free Tanner graph: dense H [m, n_vars], every check of degree >= 2.  Grown from one
    variable by hanging a new check (with 1..max_check_degree-1 new variables) on an existing variable;
    variable labels and row order are shuffled afterwards, so edge orders vary."""
    rows = []
    nv = 1
    while nv < n_vars:
        k = min(int(rng.randint(1, max_check_degree)), n_vars - nv)
        rows.append([int(rng.randint(nv))] + list(range(nv, nv + k)))
        nv += k
    perm = rng.permutation(n_vars)
    H = np.zeros((len(rows), n_vars), dtype=np.int8)
    for r, cols in enumerate(rows):
        H[r, perm[cols]] = 1
    return H[rng.permutation(len(rows))]


def is_forest(H):
    """True iff the Tanner graph of H (any non-zero = edge) has no cycle (union-find over variables + checks)."""
    H = np.asarray(H)
    m, n = H.shape
    parent = list(range(m + n))

    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a

    for r, c in zip(*np.nonzero(H)):
        a, b = find(int(c)), find(n + int(r))
        if a == b:
            return False
        parent[a] = b
    return True


def random_priors(rng, n, hard=0):
    """Non-uniform priors on both sides of 1/2; `hard` of them exactly 0 or 1 (certainty-1.0 checks, hqc.py:689)."""
    p = np.where(rng.rand(n) < 0.75, rng.uniform(0.02, 0.45, n), rng.uniform(0.55, 0.93, n))
    if hard:
        idx = rng.choice(n, hard, replace=False)
        p[idx] = rng.randint(0, 2, size=hard).astype(np.float64)
    return p


def feasible_syndromes(rng, H, probs, count):
    """Syndromes of error vectors the priors allow (e_j = p_j wherever p_j is 0 or 1)."""
    n = H.shape[1]
    e = (rng.rand(count, n) < 0.5).astype(np.uint8)
    e[:, probs == 0.0] = 0
    e[:, probs == 1.0] = 1
    return ((e.astype(np.int64) @ (np.asarray(H).T.astype(np.int64) & 1)) & 1).astype(np.uint8)


def check_binary_llr(got_llr, got_bits, exact_llr, rtol, atol, what):
    """Posteriors vs the enumerated ones: infinities in the same places with the same sign, finite values
    within |dL| <= atol + rtol * |L|; hard decisions (L <= 0 -> 1) exact wherever |L| clears the tolerance."""
    a = np.asarray(got_llr, dtype=np.float64).copy()
    b = np.asarray(exact_llr, dtype=np.float64)
    # "certain": the min-sum sweeps start their running minimum at the largest finite number (1e308 / FLT_MAX, as the
    # package does), so a certainty arrives as +-that rather than +-inf; anything beyond 1e30 is read as infinite
    with np.errstate(invalid="ignore"):
        a[np.abs(a) >= 1e30] = np.sign(a[np.abs(a) >= 1e30]) * np.inf
    inf = np.isinf(b)
    assert np.array_equal(np.isinf(a), inf) and np.array_equal(np.sign(a[inf]), np.sign(b[inf])), f"{what}: infinite posteriors differ"
    assert not np.isnan(a).any(), f"{what}: NaN posterior"
    tol = atol + rtol * np.abs(b[~inf])
    err = np.abs(a[~inf] - b[~inf])
    assert (err <= tol).all(), f"{what}: posterior off by {err.max():.3e} (worst ratio to tolerance {np.max(err / tol):.2f})"
    decided = inf | (np.abs(np.where(inf, 0, b)) > atol + rtol * np.abs(np.where(inf, 0, b)))
    want = (b <= 0).astype(np.uint8)
    assert np.array_equal(np.asarray(got_bits)[decided], want[decided]), f"{what}: hard decisions differ from the exact ones"
    return float(err.max()) if err.size else 0.0


# ---------------------------------------------------------------------------- q-ary
def qary_exact(H, llr_by_var, alphabets):
    """H dense [R, N] entries in {-1, 0, 1}; llr_by_var: list of N float arrays, llr_by_var[v][q] = cost of
    symbol value q - alphabets[v] (alphabets[v] = B of that variable, values in [-B, B]).  Enumerates the
    assignments whose every check sums to 0 OVER THE INTEGERS (decoder.rs:336-337: the last value is
    -sum of the others, not mod Q).  Returns (best assignment int [N], its cost, gap to the second best)."""
    H = np.asarray(H, dtype=np.int64)
    R, N = H.shape
    ranges = [np.arange(-b, b + 1) for b in alphabets]
    total = int(np.prod([len(r) for r in ranges], dtype=np.float64))
    assert total <= 3_000_000, total
    grids = np.meshgrid(*ranges, indexing="ij")
    X = np.stack([g.reshape(-1) for g in grids], axis=1)
    ok = ((X @ H.T) == 0).all(axis=1)
    X = X[ok]
    cost = np.zeros(X.shape[0])
    for v in range(N):
        cost += np.asarray(llr_by_var[v], dtype=np.float64)[X[:, v] + alphabets[v]]
    order = np.argsort(cost, kind="stable")
    best = X[order[0]]
    gap = cost[order[1]] - cost[order[0]] if len(order) > 1 else np.inf
    return best.astype(np.int8), float(cost[order[0]]), float(gap)


def qary_exact_special(H, llr_b, llr_s, B, BSUM):
    """DecoderSpecial's layout H = [H' | +-I_R] (decoder_special.rs:507-522): the N-R coefficient variables are
    enumerated, every sum variable follows from its check (it is that check's only other edge); an
    assignment whose sum variable leaves [-BSUM, BSUM] is invalid.  Returns (best int8 [N], cost, gap)."""
    H = np.asarray(H, dtype=np.int64)
    R, N = H.shape
    BV = N - R
    tail = H[:, BV:]
    assert (np.abs(tail) == np.eye(R, dtype=np.int64)).all(), "H must be [H' | +-I]"
    hs = np.diag(tail)
    ranges = [np.arange(-B, B + 1)] * BV
    grids = np.meshgrid(*ranges, indexing="ij")
    X = np.stack([g.reshape(-1) for g in grids], axis=1)
    S = -(X @ H[:, :BV].T) * hs[None, :]  # h_s x_s = -sum_j h_j x_j
    ok = (np.abs(S) <= BSUM).all(axis=1)
    X, S = X[ok], S[ok]
    cost = np.zeros(X.shape[0])
    for v in range(BV):
        cost += np.asarray(llr_b[v], dtype=np.float64)[X[:, v] + B]
    for r in range(R):
        cost += np.asarray(llr_s[r], dtype=np.float64)[S[:, r] + BSUM]
    order = np.argsort(cost, kind="stable")
    best = np.concatenate([X[order[0]], S[order[0]]])
    return best.astype(np.int8), float(cost[order[0]]), float(cost[order[1]] - cost[order[0]])


def random_qary_tree(rng, n_vars, max_check_degree=4, signed=True):
    """Cycle-free H with entries in {-1, 0, 1}, every check of degree >= 2 (plain Decoder)."""
    H = random_binary_tree(rng, n_vars, max_check_degree).astype(np.int8)
    if signed:
        H = H * rng.choice(np.array([-1, 1], dtype=np.int8), size=H.shape)
    return H


def random_special_tree(rng, R, coeffs_per_check, signed=True):
    """H = [H' | I_R] whose H' part is a chain of checks sharing one coefficient variable with the previous
    check (cycle-free); check r has coeffs_per_check[r] coefficient edges."""
    cols, nv = [], 0
    for r, k in enumerate(coeffs_per_check):
        if r == 0:
            cols.append(list(range(k)))
            nv = k
        else:
            cols.append([int(rng.choice(cols[r - 1]))] + list(range(nv, nv + k - 1)))
            nv += k - 1
    Hp = np.zeros((R, nv), dtype=np.int8)
    for r, cs in enumerate(cols):
        Hp[r, cs] = rng.choice(np.array([-1, 1], dtype=np.int8), size=len(cs)) if signed else 1
    return np.concatenate([Hp, np.eye(R, dtype=np.int8)], axis=1)


def pmf_to_llr64(pmf):
    """ln(max / p) in float64 -- the cost table the q-ary decoders minimise (decoder.rs:668-692)."""
    p = np.asarray(pmf, dtype=np.float64)
    with np.errstate(divide="ignore"):
        return np.log(p.max(axis=-1, keepdims=True) / p)


# ------------------------------------------------ exact inference on LARGE trees (no enumeration)
def tree_exact_binary(H, probs, synds):
    """Exact posteriors on a cycle-free Tanner graph of ANY size, by variable elimination in the log-probability domain:
    every node passes a PAIR (log weight of "0", log weight of "1") to its parent (leaves to roots), then the roots pass
    the complementary pairs back down; a check folds its children's pairs into (log weight of even parity, of odd parity)
    with logaddexp [sum-product] or max [min-sum's max-product] -- no tanh, no atanh, no LLR differences, float64, so it
    shares neither formulation nor precision with the decoders.  Agrees with `binary_exact` where both apply.
    H dense 0/1 [m, n]; probs [n]; synds uint8 [batch, m].  Returns dict(sp [batch, n], ms [batch, n])."""
    H = np.asarray(H) & 1
    m, n = H.shape
    assert is_forest(H)
    synds = np.atleast_2d(np.asarray(synds)).astype(np.int64) & 1
    B = synds.shape[0]
    p = np.asarray(probs, dtype=np.float64)
    with np.errstate(divide="ignore"):
        prior = np.stack([np.log1p(-p), np.log(p)], axis=0)  # [2, n]
    var_checks = [np.flatnonzero(H[:, v]) for v in range(n)]
    check_vars = [np.flatnonzero(H[r]) for r in range(m)]
    # root every component at its lowest-numbered variable; BFS order; isolated checks (no variable) cannot occur in a decoder
    parent_of_var = [-2] * n  # check above the variable (-1: root)
    parent_of_check = [-1] * m  # variable above the check
    order = []  # ("v", id) / ("c", id) top-down
    for root in range(n):
        if parent_of_var[root] != -2:
            continue
        parent_of_var[root] = -1
        queue = [("v", root)]
        while queue:
            kind, x = queue.pop(0)
            order.append((kind, x))
            if kind == "v":
                for c in var_checks[x]:
                    if c != parent_of_var[x]:
                        parent_of_check[c] = x
                        queue.append(("c", int(c)))
            else:
                for v in check_vars[x]:
                    if v != parent_of_check[x]:
                        parent_of_var[v] = x
                        queue.append(("v", int(v)))
    out = {}
    for name, comb in (("sp", np.logaddexp), ("ms", np.maximum)):
        def fold(a, b):  # parity pair (even, odd) of two independent groups
            with np.errstate(invalid="ignore"):
                return np.stack([comb(a[0] + b[0], a[1] + b[1]), comb(a[0] + b[1], a[1] + b[0])])
        zero = np.stack([np.zeros(B), np.full(B, -np.inf)])  # an empty group has even parity
        up_v = [None] * n  # variable -> its parent check: pair [2, B]
        up_c = [None] * m  # check -> its parent variable
        for kind, x in reversed(order):
            if kind == "v":
                w = np.repeat(prior[:, x][:, None], B, axis=1)
                for c in var_checks[x]:
                    if c != parent_of_var[x]:
                        w = w + up_c[c]
                up_v[x] = w
            else:
                acc = zero
                for v in check_vars[x]:
                    if v != parent_of_check[x]:
                        acc = fold(acc, up_v[v])
                s = synds[:, x]
                # parent value t: children parity must be s ^ t
                up_c[x] = np.stack([np.where(s == 0, acc[0], acc[1]), np.where(s == 0, acc[1], acc[0])])
        down_v = [None] * n  # parent check -> variable
        down_c = [None] * m  # parent variable -> check (the variable's belief without this check)
        post = np.zeros((2, B, n))
        for kind, x in order:
            if kind == "v":
                tot = np.repeat(prior[:, x][:, None], B, axis=1)
                if parent_of_var[x] >= 0:
                    tot = tot + down_v[x]
                kids = [c for c in var_checks[x] if c != parent_of_var[x]]
                for c in kids:
                    tot = tot + up_c[c]
                post[:, :, x] = tot
                for c in kids:  # belief without check c (recomputed as a sum: no inf - inf)
                    w = np.repeat(prior[:, x][:, None], B, axis=1)
                    if parent_of_var[x] >= 0:
                        w = w + down_v[x]
                    for c2 in kids:
                        if c2 != c:
                            w = w + up_c[c2]
                    down_c[c] = w
            else:
                kids = [v for v in check_vars[x] if v != parent_of_check[x]]
                groups = [down_c[x]] + [up_v[v] for v in kids]  # the parent variable counts as a group too
                pre = [zero]
                for g in groups:
                    pre.append(fold(pre[-1], g))
                suf = [zero]
                for g in reversed(groups):
                    suf.append(fold(suf[-1], g))
                suf = suf[::-1]  # suf[i] = fold of groups[i:]
                s = synds[:, x]
                for i, v in enumerate(kids, start=1):
                    others = fold(pre[i], suf[i + 1])  # parity pair of everybody but kid i
                    down_v[v] = np.stack([np.where(s == 0, others[0], others[1]), np.where(s == 0, others[1], others[0])])
        with np.errstate(invalid="ignore"):
            out[name] = post[0] - post[1]
    return out


def random_binary_tree_sparse(rng, n_vars, max_check_degree=5):
    """As random_binary_tree, for thousands of variables: returns (rows: list of column lists, n)."""
    rows = []
    nv = 1
    while nv < n_vars:
        k = min(int(rng.randint(1, max_check_degree)), n_vars - nv)
        rows.append([int(rng.randint(nv))] + list(range(nv, nv + k)))
        nv += k
    perm = rng.permutation(n_vars)
    rows = [sorted(int(perm[c]) for c in cs) for cs in rows]
    order = rng.permutation(len(rows))
    return [rows[i] for i in order], n_vars


def tree_exact_qary(H, llr_by_var, B):
    """Min-marginals of the q-ary cost model on a cycle-free H (entries in {-1, 0, 1}) of any size, by (min, +) elimination:
    a check folds its children's cost tables into a table over their SIGNED PARTIAL SUM (integer states, range +-k*B);
    the parent's symbol x then needs partial sum = -h_p * x.  B: one alphabet bound for all variables, or a list with
    one per variable (DecoderSpecial: B for the coefficient variables, BSUM for the row-sum variables).
    Returns minmarg float64 [N, 2B+1] (a list of rows for per-variable alphabets): minmarg[v][q] = cost of the
    cheapest valid assignment with x_v = q - B_v (inf if none).  The decoders' decision for v is argmin of that row (decoder.rs:654-657) wherever it is unique."""
    H = np.asarray(H, dtype=np.int64)
    R, N = H.shape
    assert is_forest(H)
    Bv = [int(B)] * N if np.isscalar(B) else [int(x) for x in B]  # per-variable alphabets (DecoderSpecial: B, then BSUM)
    cost = [np.asarray(llr_by_var[v], dtype=np.float64) for v in range(N)]
    var_checks = [np.flatnonzero(H[:, v]) for v in range(N)]
    check_vars = [np.flatnonzero(H[r]) for r in range(R)]
    pv, pc, order = [-2] * N, [-1] * R, []
    for root in range(N):
        if pv[root] != -2:
            continue
        pv[root] = -1
        queue = [("v", root)]
        while queue:
            kind, x = queue.pop(0)
            order.append((kind, x))
            if kind == "v":
                for c in var_checks[x]:
                    if c != pv[x]:
                        pc[c] = x
                        queue.append(("c", int(c)))
            else:
                for v in check_vars[x]:
                    if v != pc[x]:
                        pv[v] = x
                        queue.append(("v", int(v)))

    def conv(a, b):  # (min, +) convolution of two tables over integer sums, each centred (index = sum + half)
        out = np.full(a.size + b.size - 1, np.inf)
        for i, ai in enumerate(a):
            if np.isfinite(ai):
                out[i : i + b.size] = np.minimum(out[i : i + b.size], ai + b)
        return out

    def signed(tab, h):  # table of h * x given the table of x
        return tab if h > 0 else tab[::-1]

    up_v, up_c = [None] * N, [None] * R
    for kind, x in reversed(order):
        if kind == "v":
            t = cost[x].copy()
            for c in var_checks[x]:
                if c != pv[x]:
                    t = t + up_c[c]
            up_v[x] = t
        else:
            acc = np.zeros(1)
            for v in check_vars[x]:
                if v != pc[x]:
                    acc = conv(acc, signed(up_v[v], H[x, v]))
            half = (acc.size - 1) // 2
            hp = H[x, pc[x]]
            Bp = Bv[pc[x]]
            t = np.full(2 * Bp + 1, np.inf)
            for q in range(2 * Bp + 1):
                need = -hp * (q - Bp)  # children's signed sum
                if abs(need) <= half:
                    t[q] = acc[need + half]
            up_c[x] = t
    down_v, down_c = [None] * N, [None] * R
    mm = [None] * N
    for kind, x in order:
        if kind == "v":
            tot = cost[x].copy()
            if pv[x] >= 0:
                tot = tot + down_v[x]
            kids = [c for c in var_checks[x] if c != pv[x]]
            for c in kids:
                tot = tot + up_c[c]
            mm[x] = tot
            for c in kids:
                w = cost[x].copy()
                if pv[x] >= 0:
                    w = w + down_v[x]
                for c2 in kids:
                    if c2 != c:
                        w = w + up_c[c2]
                down_c[c] = w
        else:
            kids = [v for v in check_vars[x] if v != pc[x]]
            groups = [signed(down_c[x], H[x, pc[x]])] + [signed(up_v[v], H[x, v]) for v in kids]
            pre = [np.zeros(1)]
            for g in groups:
                pre.append(conv(pre[-1], g))
            suf = [np.zeros(1)]
            for g in reversed(groups):
                suf.append(conv(suf[-1], g))
            suf = suf[::-1]
            for i, v in enumerate(kids, start=1):
                others = conv(pre[i], suf[i + 1])
                half = (others.size - 1) // 2
                hv = H[x, v]
                t = np.full(2 * Bv[v] + 1, np.inf)
                for q in range(2 * Bv[v] + 1):
                    need = -hv * (q - Bv[v])
                    if abs(need) <= half:
                        t[q] = others[need + half]
                down_v[v] = t
    return np.stack(mm) if np.isscalar(B) else mm
