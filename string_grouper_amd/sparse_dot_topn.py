"""Seam b2: ``sp_matmul_topn`` / ``zip_sp_matmul_topn`` with the call signature the reference
imports from the third-party ``sparse_dot_topn`` package (string_grouper/string_grouper.py:12,
call sites :725-732, :737-743, :746), executed by the MI355X library (K3 + K4, K5).

Semantics (see include/sg_hip.h): values strictly greater than ``threshold`` are kept, at most
``top_n`` per row, chosen and -- with ``sort=True`` -- ordered by (value descending, column
ascending).  Inputs must be non-negative (TF-IDF matrices are), so "no threshold" equals 0.
"""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np
import scipy.sparse as sp

from . import _native as N


def _right_as_csr(B) -> sp.csr_matrix:
    """The reference passes ``duplicate_matrix.transpose()`` (V x n_right, CSC view of a CSR).
    The device wants the right-hand matrix itself, n_right x V in CSR: undo the transpose."""
    if sp.isspmatrix_csc(B) or (hasattr(B, "format") and B.format == "csc"):
        return B.T.tocsr()            # zero-copy for the view the reference creates
    return sp.csr_matrix(B).T.tocsr()


def _same_matrix(A: sp.csr_matrix, Bm: sp.csr_matrix) -> bool:
    """A self-join as the reference issues it: ``sp_matmul_topn(master_matrix, master_matrix.transpose(), ...)``
    (string_grouper.py:725-729) -- the transposed view shares the three arrays of the matrix.  The library then takes the
    self-join form (every pair scored once) and indexes identical rows once."""
    if Bm is A:
        return True
    return (A.shape == Bm.shape and A.dtype == Bm.dtype and A.indptr.dtype == Bm.indptr.dtype
            and A.data.ctypes.data == Bm.data.ctypes.data and A.indices.ctypes.data == Bm.indices.ctypes.data
            and A.indptr.ctypes.data == Bm.indptr.ctypes.data and A.data.shape == Bm.data.shape)


def sp_matmul_topn(A, B, top_n: int, threshold: Optional[float] = None, sort: bool = False,
                   density: Optional[float] = None, n_threads: Optional[int] = None,
                   idx_dtype=None, ctx: Optional[N.Context] = None) -> sp.csr_matrix:
    """``C = A @ B`` keeping per row the ``top_n`` largest values above ``threshold``.

    ``density`` and ``n_threads`` are accepted for signature compatibility and ignored (the
    result buffer is fixed-stride on the device; the parallelism is the GPU's)."""
    ctx = ctx or N.default_context()
    A = sp.csr_matrix(A)
    Bm = _right_as_csr(B)
    if A.shape[1] != Bm.shape[1]:
        raise ValueError(f"shapes {A.shape} and {(Bm.shape[1], Bm.shape[0])} not aligned")
    if A.dtype != Bm.dtype:
        common = np.result_type(A.dtype, Bm.dtype)
        A, Bm = A.astype(common), Bm.astype(common)
    if A.dtype not in (np.float32, np.float64):
        A, Bm = A.astype(np.float64), Bm.astype(np.float64)
    thr = 0.0 if threshold is None else max(float(threshold), 0.0)
    dA = ctx.csr_from_scipy(A)
    dB = dA if _same_matrix(A, Bm) else ctx.csr_from_scipy(Bm)
    post = ctx.postings_build(dB)
    res = ctx.spgemm_topn(dA, post, int(top_n), thr, bool(sort))
    C = res.to_scipy()
    for h in (res, post, dB, dA):
        h.free()                      # (dB may be dA: free() is idempotent)
    if idx_dtype is not None:
        C.indices = C.indices.astype(idx_dtype)
        C.indptr = C.indptr.astype(idx_dtype)
    return C


def zip_sp_matmul_topn(top_n: int, C_mats: Sequence[sp.csr_matrix], ctx: Optional[N.Context] = None) -> sp.csr_matrix:
    """Merge results of ``A @ B_i`` over column blocks ``B = [B_0 | B_1 | ...]`` (string_grouper.py:746)."""
    ctx = ctx or N.default_context()
    mats = [sp.csr_matrix(C) for C in C_mats]
    n_rows = mats[0].shape[0]
    dtype = mats[0].dtype
    offs = np.concatenate([[0], np.cumsum([m.shape[1] for m in mats])]).astype(np.int64)
    parts = []
    for m in mats:
        # fixed-stride form of a host CSR block: stride = its longest row
        cnt = np.diff(m.indptr).astype(np.int32)
        stride = int(max(1, cnt.max() if len(cnt) else 1))
        parts.append(_host_block_to_topn(ctx, m, cnt, stride, dtype))
    res = ctx.topn_zip(parts, offs[:-1], int(top_n))
    C = res.to_scipy()
    C = sp.csr_matrix((C.data, C.indices, C.indptr), shape=(n_rows, int(offs[-1])))
    res.free()
    for p in parts:
        p.free()
    return C


def _host_block_to_topn(ctx: N.Context, m: sp.csr_matrix, cnt: np.ndarray, stride: int, dtype) -> "N.TopN":
    """A host CSR block with <= stride entries per row, uploaded in the multiply's fixed-stride layout."""
    n_rows = m.shape[0]
    cols = np.zeros((n_rows, stride), np.int32)
    vals = np.zeros((n_rows, stride), dtype)
    mask = np.arange(stride, dtype=np.int32)[None, :] < cnt[:, None]
    cols[mask] = m.indices
    vals[mask] = m.data
    return ctx.topn_from_host(cols, vals, cnt, m.shape[1])
