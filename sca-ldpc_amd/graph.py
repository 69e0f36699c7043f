"""Sparse Tanner-graph container (host side).

The reference hands its decoders a *dense* parity-check matrix
(`ldpc.bp_decoder(H, ...)`, simulate/decode.py:155-161, simulate/hqc.py:680,694;
`DecoderN..(H.astype(np.int8), iters)`, simulate/decode.py:230) and the decoders
scan it into their own sparse form (mod2sparse linked lists / FxHashMap,
simulate_rs/src/decoder.rs:494-553).  Here the graph is kept sparse from the
start and never densified:

  CSR  row_ptr[m+1], col_idx[nnz] (ascending column inside a row), val[nnz]
       -- the edge id `e` IS the CSR position; check nodes stream their edges
          as one contiguous range, which is what the HIP check kernels rely on.
  CSC  col_ptr[n+1], csc_edge[nnz] (edge ids grouped by column, ascending row)
       -- variable nodes gather their edges through this permutation.

Edge traversal orders (row entries ascending column, column entries ascending
row) are the ones mod2sparse and decoder.rs:507-539 produce, so summation order
in the kernels can follow the reference.
"""
from __future__ import annotations

import numpy as np


class TannerGraph:
    __slots__ = ("m", "n", "nnz", "row_ptr", "col_idx", "val", "_csc")

    def __init__(self, m, n, rows, cols, vals=None):
        rows = np.asarray(rows, dtype=np.int64)
        cols = np.asarray(cols, dtype=np.int64)
        if rows.shape != cols.shape or rows.ndim != 1:
            raise ValueError("rows/cols must be 1-D arrays of equal length")
        if rows.size and (rows.min() < 0 or rows.max() >= m or cols.min() < 0 or cols.max() >= n):
            raise ValueError("edge index out of range")
        if vals is None:
            vals = np.ones(rows.size, dtype=np.int8)
        vals = np.asarray(vals, dtype=np.int8)
        # CSR order: by row, then column.  Callers on the attack loop's critical path (one graph per
        # decode, hqc.py:680) hand the entries over already in that order: one vectorised check
        # instead of a 200 000-entry lexsort.
        if rows.size > 1:
            dr = np.diff(rows)
            in_order = bool(((dr > 0) | ((dr == 0) & (np.diff(cols) > 0))).all())
        else:
            in_order = True
        if not in_order:
            order = np.lexsort((cols, rows))
            rows, cols, vals = rows[order], cols[order], vals[order]
            if rows.size > 1:
                dup = (rows[1:] == rows[:-1]) & (cols[1:] == cols[:-1])
                if dup.any():
                    raise ValueError("duplicate edges in parity-check matrix")
        self.m = int(m)
        self.n = int(n)
        self.nnz = int(rows.size)
        self.row_ptr = np.zeros(m + 1, dtype=np.int32)
        np.cumsum(np.bincount(rows, minlength=m), out=self.row_ptr[1:])
        self.col_idx = cols.astype(np.int32)
        self.val = vals
        self._csc = None  # built on first use: the HIP library derives its own from the CSR arrays

    def _build_csc(self):
        # CSC permutation: by column, then row (a stable sort of the CSR order by column)
        rows = np.repeat(np.arange(self.m, dtype=np.int64), np.diff(self.row_ptr))
        corder = np.argsort(self.col_idx, kind="stable")
        col_ptr = np.zeros(self.n + 1, dtype=np.int32)
        np.cumsum(np.bincount(self.col_idx, minlength=self.n), out=col_ptr[1:])
        self._csc = (col_ptr, corder.astype(np.int32), rows[corder].astype(np.int32))
        return self._csc

    @property
    def col_ptr(self):
        return (self._csc or self._build_csc())[0]

    @property
    def csc_edge(self):
        return (self._csc or self._build_csc())[1]

    @property
    def csc_row(self):
        return (self._csc or self._build_csc())[2]

    # -- constructors -------------------------------------------------------
    @classmethod
    def from_csr(cls, m, n, row_ptr, col_idx, vals=None):
        """Adopt CSR arrays as they stand (columns strictly ascending inside each row -- the
        caller's promise; the HIP library re-validates what it is handed).  No sorting, no copies
        beyond dtype conversion: the per-decode graph of the attack loop (hqc.py:680)."""
        g = cls.__new__(cls)
        g.m, g.n = int(m), int(n)
        g.row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int32)
        g.col_idx = np.ascontiguousarray(col_idx, dtype=np.int32)
        g.nnz = int(g.col_idx.size)
        if g.row_ptr.shape != (g.m + 1,) or g.row_ptr[0] != 0 or g.row_ptr[-1] != g.nnz:
            raise ValueError("row_ptr does not span [0, nnz]")
        g.val = np.ones(g.nnz, dtype=np.int8) if vals is None else np.ascontiguousarray(vals, dtype=np.int8)
        g._csc = None
        return g

    @classmethod
    def from_dense(cls, H):
        H = np.asarray(H)
        if H.ndim != 2:
            raise ValueError("parity-check matrix must be 2-D")
        r, c = np.nonzero(H)
        return cls(H.shape[0], H.shape[1], r, c, H[r, c])

    @classmethod
    def from_coo(cls, d):
        """From the JSON COO form used by tests/golden."""
        return cls(d["shape"][0], d["shape"][1], d["rows"], d["cols"], d.get("vals"))

    @classmethod
    def from_row_supports(cls, supports, n):
        """supports: list of 1-D index arrays, one per check row."""
        lens = [len(s) for s in supports]
        rows = np.repeat(np.arange(len(supports)), lens)
        cols = np.concatenate([np.asarray(s, dtype=np.int64) for s in supports]) if supports else np.zeros(0, np.int64)
        return cls(len(supports), n, rows, cols)

    @classmethod
    def coerce(cls, H):
        if isinstance(H, cls):
            return H
        if hasattr(H, "tocoo"):  # scipy sparse
            c = H.tocoo()
            return cls(c.shape[0], c.shape[1], c.row, c.col, c.data)
        return cls.from_dense(H)

    # -- views --------------------------------------------------------------
    def to_dense(self, dtype=int):
        H = np.zeros((self.m, self.n), dtype=dtype)
        rows = np.repeat(np.arange(self.m), np.diff(self.row_ptr))
        H[rows, self.col_idx] = self.val
        return H

    def with_identity(self):
        """[H | I_m] -- what simulate/hqc.py:680 concatenates densely."""
        rows = np.repeat(np.arange(self.m), np.diff(self.row_ptr))
        r = np.concatenate([rows, np.arange(self.m)])
        c = np.concatenate([self.col_idx.astype(np.int64), self.n + np.arange(self.m)])
        v = np.concatenate([self.val, np.ones(self.m, dtype=np.int8)])
        return TannerGraph(self.m, self.n + self.m, r, c, v)

    def row_degrees(self):
        return np.diff(self.row_ptr)

    def col_degrees(self):
        return np.diff(self.col_ptr)

    def syndrome(self, x):
        """(H @ x) % 2 for a 0/1 vector or a [batch, n] array, sparse (decode.py:168)."""
        x = np.asarray(x)
        rows = np.repeat(np.arange(self.m), np.diff(self.row_ptr))
        if x.ndim == 1:
            out = np.zeros(self.m, dtype=np.int64)
            np.add.at(out, rows, x[self.col_idx].astype(np.int64) & 1)
            return (out & 1).astype(np.uint8)
        # batches: one sparse x dense product, 256 codewords at a time (a [batch, nnz] gather would need
        # batch * nnz * 8 bytes -- 6.7 GB for 4096 codewords on the HQC-128 bench graph)
        try:
            import scipy.sparse as sp

            A = sp.csr_matrix((np.ones(self.nnz, dtype=np.int32), self.col_idx, self.row_ptr), shape=(self.m, self.n))
            out = np.empty((x.shape[0], self.m), dtype=np.uint8)
            for b0 in range(0, x.shape[0], 256):
                xb = (x[b0 : b0 + 256].astype(np.int32) & 1).T
                out[b0 : b0 + 256] = (np.asarray(A @ xb).T & 1).astype(np.uint8)
            return out
        except ImportError:
            pass
        out = np.zeros((x.shape[0], self.m), dtype=np.int64)
        for b in range(x.shape[0]):
            out[b] = np.bincount(rows, weights=(x[b, self.col_idx].astype(np.int64) & 1), minlength=self.m).astype(np.int64)
        return (out & 1).astype(np.uint8)

    def __repr__(self):
        return f"TannerGraph(m={self.m}, n={self.n}, nnz={self.nnz})"
