"""`ldpc.codes` subset used by the reference (main.py:42,273; decode.py:140)."""
import numpy as np


def rep_code(distance):
    """(distance-1) x distance parity-check matrix of the repetition code:
    row i has ones in columns i and i+1.  Dense, as the reference's driver expects
    (`H.shape`, `H @ error % 2`, decode.py:151,168)."""
    H = np.zeros((distance - 1, distance), dtype=int)
    i = np.arange(distance - 1)
    H[i, i] = 1
    H[i, i + 1] = 1
    return H
