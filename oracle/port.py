"""ctypes wrapper around oracle/sdtn_port.c -- TEST INFRASTRUCTURE ONLY (see oracle.py header).

``sp_matmul_topn_port`` has the signature of sparse_dot_topn.sp_matmul_topn (call sites
string_grouper/string_grouper.py:725-732, :737-743) and is the CPU baseline ("kind": "port")
timed by bench.py."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np
import scipy.sparse as sp

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libsdtn_port.so")
    src = os.path.join(_HERE, "sdtn_port.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libsdtn_port.so"])
    return so


def _lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        _LIB.sdtn_count_macs.restype = ctypes.c_int64
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _as_bt_csr(B):
    """B is V x nR (what the reference passes as duplicate_matrix.transpose()); return CSR."""
    Bt = sp.csr_matrix(B)
    Bt.sort_indices()
    return Bt


def sp_matmul_topn_port(A, B, top_n, threshold=0.0, sort=True, n_threads=None, tie_rule=0):
    """tie_rule 0: the canonical order this build defines (score descending, column ascending); 1: the arrival-order
    variant of sdtn_port.c (what a bounded heap that only replaces on strictly greater values would keep)."""
    A = sp.csr_matrix(A)
    if not A.has_sorted_indices:
        A = A.sorted_indices()
    Bt = _as_bt_csr(B)
    assert A.shape[1] == Bt.shape[0]
    dtype = A.dtype
    assert dtype in (np.float32, np.float64) and Bt.dtype == dtype
    nL, nR = A.shape[0], Bt.shape[1]
    top_n = int(max(1, min(top_n, max(nR, 1))))
    a_ip = A.indptr.astype(np.int64); a_ix = A.indices.astype(np.int32); a_d = np.ascontiguousarray(A.data)
    b_ip = Bt.indptr.astype(np.int64); b_ix = Bt.indices.astype(np.int32); b_d = np.ascontiguousarray(Bt.data)
    oc = np.empty(nL * top_n, np.int32); ov = np.empty(nL * top_n, dtype); cnt = np.zeros(nL, np.int32)
    lib = _lib()
    fn = lib.sdtn_sp_matmul_topn_f32 if dtype == np.float32 else lib.sdtn_sp_matmul_topn_f64
    thr = ctypes.c_float(float(np.float32(threshold))) if dtype == np.float32 else ctypes.c_double(float(threshold))
    rc = fn(ctypes.c_int64(nL), ctypes.c_int64(nR), _p(a_ip), _p(a_ix), _p(a_d), _p(b_ip), _p(b_ix), _p(b_d),
            ctypes.c_int32(top_n), thr, ctypes.c_int32(1 if sort else 0),
            ctypes.c_int32(int(n_threads) if n_threads else 1), _p(oc), _p(ov), _p(cnt), ctypes.c_int32(int(tie_rule)))
    if rc != 0:
        raise MemoryError("sdtn_port: allocation failed")
    return fixed_stride_to_csr(oc, ov, cnt, top_n, (nL, nR))


def fixed_stride_to_csr(cols, vals, cnt, stride, shape):
    """[nL x stride] (col, val) + count per row  ->  scipy CSR keeping the within-row order."""
    nL = shape[0]
    indptr = np.zeros(nL + 1, np.int64)
    np.cumsum(cnt, out=indptr[1:])
    mask = np.arange(stride, dtype=np.int32)[None, :] < cnt[:, None]
    indices = cols.reshape(nL, stride)[mask]
    data = vals.reshape(nL, stride)[mask]
    idx_dtype = np.int32 if indptr[-1] < 2**31 else np.int64
    return sp.csr_matrix((data, indices.astype(np.int32), indptr.astype(idx_dtype)), shape=shape)


def count_macs(A, B):
    A = sp.csr_matrix(A); Bt = _as_bt_csr(B)
    a_ip = A.indptr.astype(np.int64); a_ix = A.indices.astype(np.int32); b_ip = Bt.indptr.astype(np.int64)
    return int(_lib().sdtn_count_macs(ctypes.c_int64(A.shape[0]), _p(a_ip), _p(a_ix), _p(b_ip)))
