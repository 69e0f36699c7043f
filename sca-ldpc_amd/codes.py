"""Parity-check constructors, restated sparse.

Each constructor consumes its `np.random.RandomState` exactly as the
reference's does, so the same seed yields the same code (pinned in
tests/test_codes.py against matrices minted from the reference itself), but
none of them allocates a dense N x N circulant (the reference's HQC path builds
`scipy.linalg.circulant(first_row)`: 17669^2 int64 = 2.5 GB, make_code.py:242-243).

Reference sources: simulate/make_code.py:12-29 (fixed_weight_vec), :97-123 (QC),
:126-188 (regular Gallager), :191-217 (+identity), :220-244 (HQC circulant),
:50-94 (q-ary QC); simulate/distance_spectrum.py:47-87; `ldpc.codes.rep_code`
(third-party, main.py:42,273).
"""
from __future__ import annotations

from math import ceil

import numpy as np

from .graph import TannerGraph


def make_random_state(seed):
    """simulate/utils.py:14-42."""
    import numbers

    if seed is None or seed is np.random:
        return np.random.mtrand._rand
    if isinstance(seed, numbers.Integral):
        return np.random.RandomState(seed)
    if isinstance(seed, np.random.RandomState):
        return seed
    raise ValueError("%r cannot be used to seed a numpy.random.RandomState instance" % seed)


# --------------------------------------------------------------------------
# binary codes
# --------------------------------------------------------------------------
def rep_code_graph(n: int) -> TannerGraph:
    """(n-1) x n bidiagonal repetition-code checks: row i = {i, i+1}."""
    i = np.arange(n - 1)
    return TannerGraph(n - 1, n, np.concatenate([i, i]), np.concatenate([i, i + 1]))


def fixed_weight_support(size: int, samplings: int, rng) -> np.ndarray:
    """Support of make_code.fixed_weight_vec (note randint(0, size-1): the last
    index is never drawn -- reference quirk kept for seed parity)."""
    chosen = np.zeros(size, dtype=bool)
    w = 0
    while w < samplings:
        i = rng.randint(0, size - 1)
        if not chosen[i]:
            chosen[i] = True
            w += 1
    return np.nonzero(chosen)[0]


def circulant_row_support(first_col_support, N: int, i: int) -> np.ndarray:
    """Support of row i of scipy.linalg.circulant(c): C[i, j] = c[(i - j) mod N]."""
    return np.sort((i - np.asarray(first_col_support, dtype=np.int64)) % N)


def circulant_graph(first_col_support, N: int, rows=None) -> TannerGraph:
    """Rows `rows` (default all) of circulant(c) as a sparse graph."""
    k = np.asarray(first_col_support, dtype=np.int64)
    rows = np.arange(N, dtype=np.int64) if rows is None else np.asarray(rows, dtype=np.int64)
    cols = (rows[:, None] - k[None, :]) % N
    r = np.repeat(np.arange(rows.size), k.size)
    return TannerGraph(rows.size, N, r, cols.ravel())


def make_qc_parity_check_graph(block_len, column_weight, num_blocks, rng) -> TannerGraph:
    """[H_0 | ... | H_{nb-1} | I], H_i circulant (make_code.py:97-123)."""
    rr, cc = [], []
    for b in range(num_blocks):
        sup = fixed_weight_support(block_len, column_weight, rng)
        g = circulant_graph(sup, block_len)
        rows = np.repeat(np.arange(block_len), np.diff(g.row_ptr))
        rr.append(rows)
        cc.append(g.col_idx.astype(np.int64) + b * block_len)
    rr.append(np.arange(block_len))
    cc.append(num_blocks * block_len + np.arange(block_len))
    return TannerGraph(block_len, (num_blocks + 1) * block_len, np.concatenate(rr), np.concatenate(cc))


def make_regular_ldpc_graph(k, r, column_weight, row_weight, rng) -> TannerGraph:
    """Gallager-style regular code (make_code.py:126-188).

    Block 0 has row i covering columns [i*rw, (i+1)*rw); each further block is a
    column permutation of block 0 drawn as `rng.permutation(block.T).T`, i.e. new
    column j is old column perm[j] where perm is the row shuffle of block.T.
    """
    if column_weight <= 1:
        raise ValueError("column_weight must be at least 2.")
    if row_weight < column_weight:
        raise ValueError("row_weight must be greater than or equal column_weight.")
    if k % row_weight:
        raise ValueError("row_weight must divide n for a regular LDPC matrix H.")
    if r != (k * column_weight) // row_weight:
        raise ValueError("r must follow '(k * column_weight) // row_weight' for the parity check matrix to be regular")
    block_size = r // column_weight
    col = np.arange(block_size * row_weight)
    rows = [col // row_weight]
    cols = [col]
    base_row_of_col = np.full(k, -1, dtype=np.int64)
    base_row_of_col[col] = col // row_weight
    for i in range(1, column_weight):
        # RandomState.permutation on a 2-D array shuffles its first axis with the
        # same draws as shuffling arange(k): new[j] = old[perm[j]]
        perm = rng.permutation(k)
        src_row = base_row_of_col[perm]  # row (within block) of new column j
        j = np.nonzero(src_row >= 0)[0]
        rows.append(src_row[j] + i * block_size)
        cols.append(j)
    return TannerGraph(r, k, np.concatenate(rows), np.concatenate(cols))


def make_regular_ldpc_identity_graph(k, r, column_weight, row_weight, rng) -> TannerGraph:
    """[H | I_r] (make_code.py:191-217)."""
    return make_regular_ldpc_graph(k, r, column_weight, row_weight, rng).with_identity()


# --------------------------------------------------------------------------
# HQC: circulant rows with distance-spectrum multiplicity <= limit
# --------------------------------------------------------------------------
def gen_support_ds_multiplicity(length: int, weight: int, max_multiplicity: int, rng) -> np.ndarray:
    """Support of distance_spectrum.gen_array_ds_multiplicity (:47-87), O(weight) per
    candidate and without the reference's per-candidate gc.collect().

    The reference draws `rng.choice(length, size=length, replace=False)` once and
    greedily accepts a candidate position iff adding it keeps every circular
    distance's multiplicity <= max_multiplicity (check_ds_addition_limit, :25-44;
    note that helper bumps the counter per existing one, so a candidate that hits
    the same distance twice is tested against the limit cumulatively)."""
    choices = rng.choice(length, size=length, replace=False)
    ds = np.zeros(length // 2 + 1, dtype=np.int64)
    support = [int(choices[0])]
    if weight <= 1:
        return np.array(sorted(support), dtype=np.int64)
    for nxt in choices[1:]:
        nxt = int(nxt)
        d = np.abs(nxt - np.asarray(support, dtype=np.int64))
        d = np.minimum(d, length - d)
        cnt = np.bincount(d, minlength=ds.size)
        if ((ds + cnt)[d] <= max_multiplicity).all():
            ds += cnt
            support.append(nxt)
            if len(support) >= weight:
                return np.array(sorted(support), dtype=np.int64)
    raise Exception(f"Failed to find a random array with more than {len(support)} number of set positions")


def calc_ds(support, length: int) -> np.ndarray:
    """distance_spectrum.calc_ds (:5-22) on a support."""
    s = np.sort(np.asarray(support, dtype=np.int64))
    out = np.zeros(length // 2 + 1, dtype=np.int64)
    for i in range(s.size):
        d = s[i + 1 :] - s[i]
        d = np.minimum(d, length - d)
        np.add.at(out, d, 1)
    return out


def make_random_ldpc_first_row(n: int, weight: int, rng) -> np.ndarray:
    """First *column* support c of the HQC circulant H0 = circulant(c)
    (make_code.py:220-244)."""
    return gen_support_ds_multiplicity(n, weight, 1, rng)


def hqc_check_graph(first_support, N: int, rows) -> TannerGraph:
    """Hin = the given rows of circulant(first_support): one row per oracle answer
    (hqc.py:885-908 stacks Hgen[bit_n])."""
    return circulant_graph(first_support, N, rows)


# --------------------------------------------------------------------------
# q-ary QC (Kyber-shaped) code, entries in {-1, 0, +1}
# --------------------------------------------------------------------------
def _circular_qary_block(block_len, column_weight, rng):
    """make_code.circular_qary_parity_check_block (:50-68) as COO."""
    nz = set()
    while len(nz) < column_weight:
        i = rng.randint(0, block_len - 1)
        if i not in nz:
            nz.add(i)
    idx = list(nz)
    val = [1 if i == 0 else -1 for i in idx]
    rr, cc, vv = [], [], []
    for i in range(block_len):
        for j in range(column_weight):
            rr.append(i)
            cc.append(idx[j])
            vv.append(val[j])
            idx[j] += 1
            if idx[j] == block_len:
                idx[j] = 0
                val[j] = -val[j]
    return np.array(rr), np.array(cc), np.array(vv)


def make_qary_qc_graph(block_len, sum_weight, num_blocks, rng, check_blocks=1) -> TannerGraph:
    """[M | I] with M a check_blocks x num_blocks grid of signed circulant-like
    blocks (make_code.py:72-94)."""
    column_weight = ceil(sum_weight / num_blocks)
    if sum_weight % num_blocks != 0:
        raise NotImplementedError()
    rr, cc, vv = [], [], []
    for cb in range(check_blocks):
        for b in range(num_blocks):
            r, c, v = _circular_qary_block(block_len, column_weight, rng)
            rr.append(r + cb * block_len)
            cc.append(c + b * block_len)
            vv.append(v)
    R = block_len * check_blocks
    rr.append(np.arange(R))
    cc.append(block_len * num_blocks + np.arange(R))
    vv.append(np.ones(R, dtype=np.int64))
    return TannerGraph(R, block_len * num_blocks + R, np.concatenate(rr), np.concatenate(cc), np.concatenate(vv))


# --------------------------------------------------------------------------
# BASELINE benchmark graphs
# --------------------------------------------------------------------------
HQC_PARAMS = {
    # name: (N, omega)  -- HQC round-3 constants; N/omega come from liboqs at run
    # time in the reference (simulate_rs/src/hqc.rs:35-46)
    "hqc128": (17669, 66),
    "hqc192": (35851, 100),
    "hqc256": (57637, 131),
}
BENCH_R = {"hqc128": 4000, "hqc192": 8000, "hqc256": 12000}


def hqc_bench_graph(name: str, first_support, R=None, row_seed: int = 1):
    """H = [Hin | I_R]: Hin = R rows of circulant(first_support), the rows being the
    first R entries of RandomState(row_seed).permutation(N).  Returns (H, Hin, rows)."""
    N, _ = HQC_PARAMS[name]
    R = BENCH_R[name] if R is None else R
    rows = np.random.RandomState(row_seed).permutation(N)[:R]
    Hin = hqc_check_graph(first_support, N, rows)
    return Hin.with_identity(), Hin, rows
