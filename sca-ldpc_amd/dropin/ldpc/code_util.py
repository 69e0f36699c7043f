"""`ldpc.code_util.get_code_parameters`, which the reference calls only to log the
result (simulate/hqc.py:1267-1270).  Returns (n, k, d, max column weight, max row
weight); d is found by enumeration and is None when k > 20."""
import itertools

import numpy as np


def _rank_gf2(H):
    A = (np.asarray(H) % 2).astype(np.uint8).copy()
    r = 0
    rows, cols = A.shape
    for c in range(cols):
        piv = None
        for i in range(r, rows):
            if A[i, c]:
                piv = i
                break
        if piv is None:
            continue
        A[[r, piv]] = A[[piv, r]]
        for i in range(rows):
            if i != r and A[i, c]:
                A[i] ^= A[r]
        r += 1
        if r == rows:
            break
    return r


def get_code_parameters(H):
    H = np.asarray(H)
    n = H.shape[1]
    k = n - _rank_gf2(H)
    d = None
    if 0 < k <= 20 and n <= 64:
        Hb = (H % 2).astype(np.uint8)
        for w in range(1, n + 1):
            for sup in itertools.combinations(range(n), w):
                if not (Hb[:, list(sup)].sum(axis=1) % 2).any():
                    d = w
                    break
            if d is not None:
                break
    nz = H != 0
    return n, k, d, int(nz.sum(axis=0).max()), int(nz.sum(axis=1).max())
