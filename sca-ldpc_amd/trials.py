"""Synthetic HQC attack trials (host side): the inputs `hqc.decode()` assembles
(simulate/hqc.py:678-705), generated without the liboqs oracle.

Trial i (global index, independent of how trials are sharded over GPUs):
    rng   = RandomState(base_seed + i)
    y     = omega distinct positions            (the secret's support)
    c     = Hin y mod 2, each bit flipped with probability eps   (noisy oracle answers)
    msg   = [0]*N ++ c                          (hqc.py:703-705)
    prior = [omega/N]*N ++ [eps]*R              (hqc.py:684-690 with certainty 1-eps)
"""
from __future__ import annotations

import numpy as np


def hqc_trials(Hin, omega, eps, count, base_seed=2, first_index=0):
    """Returns (msg uint8 [count, N+R], y_support int32 [count, omega])."""
    N, R = Hin.n, Hin.m
    msg = np.zeros((count, N + R), dtype=np.uint8)
    ys = np.zeros((count, omega), dtype=np.int32)
    col_ptr, csc_row = Hin.col_ptr, Hin.csc_row
    for i in range(count):
        rng = np.random.RandomState(base_seed + first_index + i)
        y = rng.choice(N, omega, replace=False)
        ys[i] = y
        hit = np.concatenate([csc_row[col_ptr[j] : col_ptr[j + 1]] for j in y]) if omega else np.zeros(0, np.int64)
        c = (np.bincount(hit, minlength=R) & 1).astype(np.uint8)
        if eps > 0:
            c ^= (rng.rand(R) < eps).astype(np.uint8)
        msg[i, N:] = c
    return msg, ys


def hqc_priors(N, R, omega, eps):
    return np.concatenate([np.full(N, omega / N, dtype=np.float64), np.full(R, float(eps), dtype=np.float64)])


def success(bits, ys, N):
    """hqc.py:742-749: decoded[:N] must equal the indicator of y."""
    truth = np.zeros((bits.shape[0], N), dtype=np.uint8)
    np.put_along_axis(truth, ys.astype(np.int64), 1, axis=1)
    return (bits[:, :N] == truth).all(axis=1)
