"""ctypes binding of libsg_hip.so (C ABI declared in include/sg_hip.h).

There is no CPU fallback: importing this module without the built library, or creating a
context without an MI355X, raises.  The binding passes plain pointers and sizes only."""
from __future__ import annotations

import ctypes as C
import os
import threading
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SG_HIP_LIB: diagnostic hook -- another build of the SAME library (e.g. the -DSG_WATCHDOG build scripts/ make on the
# GPU box); never a different implementation
LIB_PATH = os.environ.get("SG_HIP_LIB") or os.path.join(_HERE, "libsg_hip.so")

SG_OK, SG_ERR_BADARG, SG_ERR_OOM, SG_ERR_OVERFLOW, SG_ERR_HIP, SG_ERR_NODEVICE, SG_ERR_UNSUPPORTED = range(7)
SG_F32, SG_F64 = 0, 1
ABI_VERSION = 3          # include/sg_hip.h: SG_ABI_VERSION
SG_K_TOKENIZE, SG_K_WEIGHT, SG_K_POSTINGS, SG_K_SPGEMM, SG_K_ZIP, SG_K_VOCAB, SG_K_SPGEMM_KERNEL, SG_K_COUNT = range(8)
KERNEL_NAMES = ("tokenize", "weight", "postings", "spgemm_topn", "zip", "vocab", "spgemm_kernel")


class SgVecParams(C.Structure):
    _fields_ = [("ngram_size", C.c_int32), ("ascii_lower", C.c_int32), ("dtype", C.c_int32),
                ("reserved", C.c_int32), ("delete_table", C.c_uint8 * 128)]


class SgStats(C.Structure):
    _fields_ = [("ms", C.c_float * SG_K_COUNT), ("macs", C.c_int64), ("spgemm_bytes", C.c_int64),
                ("out_nnz", C.c_int64), ("prune_rows", C.c_int64), ("prune_postings", C.c_int64),
                ("prune_survivors", C.c_int64), ("exact_rows", C.c_int64), ("prune_bytes", C.c_int64),
                ("prune_symmetric", C.c_int64), ("prune_scored", C.c_int64)]


# every symbol include/sg_hip.h declares: name -> (restype, argtypes)
_P = C.c_void_p
_PP = C.POINTER(C.c_void_p)
ABI = {
    "sg_last_error": (C.c_char_p, []),
    "sg_abi_version": (C.c_int, []),
    "sg_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "sg_ctx_create": (C.c_int, [C.c_int, _P, _PP]),
    "sg_ctx_set_option": (C.c_int, [_P, C.c_char_p, C.c_char_p]),
    "sg_ctx_reset_options": (C.c_int, [_P]),
    "sg_ctx_options": (C.c_int, [_P, C.c_char_p, C.c_int64]),
    "sg_ctx_destroy": (C.c_int, [_P]),
    "sg_ctx_sync": (C.c_int, [_P]),
    "sg_ctx_trim": (C.c_int, [_P]),
    "sg_strings_from_host": (C.c_int, [_P, _P, _P, C.c_int64, _PP]),
    "sg_strings_from_device": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int64, _PP]),
    "sg_strings_from_host_symbols": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int32, _PP]),
    "sg_strings_set_prelowered": (C.c_int, [_P, C.c_int32]),
    "sg_strings_free": (C.c_int, [_P]),
    "sg_vec_fit": (C.c_int, [_P, _PP, C.c_int32, C.POINTER(SgVecParams), _PP]),
    "sg_vec_fit_begin": (C.c_int, [_P, _PP, C.c_int32, C.POINTER(SgVecParams), _PP]),
    "sg_vocab_df_table": (C.c_int, [_P, _PP, C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    "sg_vec_fit_end": (C.c_int, [_P, _P, C.c_int64]),
    "sg_vocab_size": (C.c_int, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "sg_vocab_coding": (C.c_int, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "sg_vocab_to_host": (C.c_int, [_P, _P, _P, _P]),
    "sg_vocab_byte_alphabet": (C.c_int, [_P, _P, C.POINTER(C.c_int32)]),
    "sg_vocab_set_idf": (C.c_int, [_P, _P, _P, C.c_int32]),
    "sg_ctx_put_idf_table": (C.c_int, [_P, C.c_int64, C.c_int32, _P]),
    "sg_vocab_apply_idf_table": (C.c_int, [_P, _P, C.POINTER(C.c_int32)]),
    "sg_vocab_free": (C.c_int, [_P]),
    "sg_vec_transform": (C.c_int, [_P, _P, _P, _PP]),
    "sg_csr_from_host": (C.c_int, [_P, C.c_int64, C.c_int64, _P, _P, _P, C.c_int32, _PP]),
    "sg_csr_from_device": (C.c_int, [_P, C.c_int64, C.c_int64, C.c_int64, _P, _P, _P, C.c_int32, _PP]),
    "sg_csr_dims": (C.c_int, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                              C.POINTER(C.c_int32)]),
    "sg_csr_device_ptrs": (C.c_int, [_P, _PP, _PP, _PP]),
    "sg_csr_to_host": (C.c_int, [_P, _P, _P, _P, _P]),
    "sg_csr_row_block": (C.c_int, [_P, _P, C.c_int64, C.c_int64, _PP]),
    "sg_csr_free": (C.c_int, [_P]),
    "sg_postings_build": (C.c_int, [_P, _P, C.c_int32, _PP]),
    "sg_postings_build_flags": (C.c_int, [_P, _P, C.c_int32, C.c_int32, _PP]),
    "sg_postings_free": (C.c_int, [_P]),
    "sg_spgemm_topn": (C.c_int, [_P, _P, _P, C.c_int32, C.c_double, C.c_int32, _PP]),
    "sg_topn_dims": (C.c_int, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                               C.POINTER(C.c_int64)]),
    "sg_topn_device_ptrs": (C.c_int, [_P, _PP, _PP, _PP]),
    "sg_topn_to_host": (C.c_int, [_P, _P, _P, _P, _P]),
    "sg_topn_counts_to_host": (C.c_int, [_P, _P, _P]),
    "sg_topn_from_host": (C.c_int, [_P, C.c_int64, C.c_int64, C.c_int32, C.c_int32, _P, _P, _P, _PP]),
    "sg_topn_from_device": (C.c_int, [_P, C.c_int64, C.c_int64, C.c_int32, C.c_int32, _P, _P, _P, _PP]),
    "sg_topn_zip": (C.c_int, [_P, _PP, _P, C.c_int32, C.c_int32, _PP]),
    "sg_topn_free": (C.c_int, [_P]),
    "sg_sp_matmul_topn_host": (C.c_int, [_P, C.c_int64, C.c_int64, C.c_int64, _P, _P, _P, _P, _P, _P, C.c_int32,
                                         C.c_int32, C.c_double, C.c_int32, _P, _P, _P]),
    "sg_matchlist_build": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, _PP]),
    "sg_matchlist_dims": (C.c_int, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    "sg_matchlist_to_host": (C.c_int, [_P, _P, _P, _P, _P]),
    "sg_matchlist_free": (C.c_int, [_P]),
    "sg_matchlist_best_master": (C.c_int, [_P, _P, _P]),
    "sg_matchlist_group_reps": (C.c_int, [_P, _P, C.c_int32, _P]),
    "sg_row_costs": (C.c_int, [_P, _P, _P, _P]),
    "sg_selfjoin_range": (C.c_int, [_P, _P, _P, C.c_int32, C.c_double, C.c_int64, C.c_int64, _P, _P, _P, _P, _P, C.c_int64]),
    "sg_selfjoin_merge": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_int32, C.c_int64, C.c_int64, C.c_int64]),
    "sg_postings_permutation": (C.c_int, [_P, _PP, _PP]),
    "sg_postings_rows": (C.c_int, [_P, _P, _P, _PP]),
    "sg_postings_bytes": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "sg_topn_expand_groups": (C.c_int, [_P, _P, _P, _P, C.c_int64, _PP]),
    "sg_topn_expand_range": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int64, _PP, _PP, _P, C.c_int64]),
    "sg_device_free": (C.c_int, [_P, _P]),
    "sg_csr_rowwise_dot": (C.c_int, [_P, _P, _P, _P]),
    "sg_ctx_stats": (C.c_int, [_P, C.POINTER(SgStats)]),
}

_lib = None
_lock = threading.Lock()


class SgHipError(RuntimeError):
    pass


def lib():
    """Load libsg_hip.so; raise (never fall back) if it has not been built."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise ImportError(
                    f"{LIB_PATH} is missing: build the HIP extension first "
                    f"(python -c 'import __graft_entry__ as g; g.build()' or make -C string_grouper_amd/csrc). "
                    f"string_grouper_amd has no CPU fallback.")
            # When PyTorch is present it must load ITS bundled HIP runtime first: both runtimes carry the
            # SONAME libamdhip64.so.7, the first one loaded serves the whole process, and PyTorch initialised
            # on top of a different runtime than it was built for intermittently reports "No HIP GPUs".
            try:
                import torch  # noqa: F401
            except Exception:
                pass
            handle = C.CDLL(LIB_PATH)
            for name, (res, args) in ABI.items():
                fn = getattr(handle, name)       # AttributeError if a declared symbol is not exported
                fn.restype = res
                fn.argtypes = args
            got = handle.sg_abi_version()
            if got != ABI_VERSION:
                raise ImportError(f"{LIB_PATH} has ABI version {got}, this binding was written for {ABI_VERSION} "
                                  f"(include/sg_hip.h: SG_ABI_VERSION): rebuild the library (make -C string_grouper_amd/csrc)")
            _lib = handle
    return _lib


def check(status: int):
    if status == SG_OK:
        return
    msg = lib().sg_last_error().decode("utf-8", "replace")
    if status == SG_ERR_OVERFLOW:
        raise OverflowError(msg)       # the one exception the reference's fit() handles (string_grouper.py:400)
    if status == SG_ERR_OOM:
        raise MemoryError(msg)
    if status == SG_ERR_BADARG:
        raise ValueError(msg)
    if status == SG_ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    raise SgHipError(f"libsg_hip status {status}: {msg}")


def device_count() -> int:
    n = C.c_int(0)
    check(lib().sg_device_count(C.byref(n)))
    return n.value


def np_dtype_code(dtype) -> int:
    dt = np.dtype(dtype)
    if dt == np.float32:
        return SG_F32
    if dt == np.float64:
        return SG_F64
    raise ValueError(f"unsupported value dtype {dt}; only float32 and float64")


def code_np_dtype(code: int):
    return np.float64 if code == SG_F64 else np.float32


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class _Handle:
    _free = None

    def __init__(self, ctx: "Context", handle):
        self.ctx = ctx
        self.h = handle

    def free(self):
        if self.h is not None and self._free is not None:
            getattr(lib(), self._free)(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Strings(_Handle):
    _free = "sg_strings_free"
    n = 0


class Vocab(_Handle):
    _free = "sg_vocab_free"


class Postings(_Handle):
    _free = "sg_postings_free"


class Csr(_Handle):
    _free = "sg_csr_free"
    parent = None  # keeps the parent of a row-block view alive

    def dims(self):
        r, c, z, d = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int32()
        check(lib().sg_csr_dims(self.h, C.byref(r), C.byref(c), C.byref(z), C.byref(d)))
        return r.value, c.value, z.value, d.value

    def to_scipy(self):
        import scipy.sparse as sp
        r, c, nnz, d = self.dims()
        indptr = np.empty(r + 1, np.int64)
        indices = np.empty(max(nnz, 1), np.int32)
        data = np.empty(max(nnz, 1), code_np_dtype(d))
        check(lib().sg_csr_to_host(self.ctx.h, self.h, _ptr(indptr), _ptr(indices), _ptr(data)))
        idx_dtype = np.int32 if nnz < 2 ** 31 else np.int64
        m = sp.csr_matrix((data[:nnz], indices[:nnz], indptr.astype(idx_dtype)), shape=(r, c))
        m.has_sorted_indices = True
        return m

    def row_block(self, r0: int, r1: int) -> "Csr":
        out = C.c_void_p()
        check(lib().sg_csr_row_block(self.ctx.h, self.h, r0, r1, C.byref(out)))
        v = Csr(self.ctx, out)
        v.parent = self
        return v


class TopN(_Handle):
    _free = "sg_topn_free"

    def dims(self):
        r, s, d, c = C.c_int64(), C.c_int32(), C.c_int32(), C.c_int64()
        check(lib().sg_topn_dims(self.h, C.byref(r), C.byref(s), C.byref(d), C.byref(c)))
        return r.value, s.value, d.value, c.value

    def to_host(self):
        r, s, d, c = self.dims()
        cols = np.empty(max(r * s, 1), np.int32)
        vals = np.empty(max(r * s, 1), code_np_dtype(d))
        cnt = np.zeros(max(r, 1), np.int32)
        check(lib().sg_topn_to_host(self.ctx.h, self.h, _ptr(cols), _ptr(vals), _ptr(cnt)))
        return cols[:r * s].reshape(r, s), vals[:r * s].reshape(r, s), cnt[:r]

    def counts(self) -> np.ndarray:
        r = self.dims()[0]
        cnt = np.zeros(max(r, 1), np.int32)
        check(lib().sg_topn_counts_to_host(self.ctx.h, self.h, _ptr(cnt)))
        return cnt[:r]

    def to_scipy(self):
        """CSR with the within-row order of the device result (score desc / col asc when sort=True)."""
        import scipy.sparse as sp
        r, s, d, c = self.dims()
        cols, vals, cnt = self.to_host()
        indptr = np.zeros(r + 1, np.int64)
        np.cumsum(cnt, out=indptr[1:])
        mask = np.arange(s, dtype=np.int32)[None, :] < cnt[:, None]
        idx_dtype = np.int32 if indptr[-1] < 2 ** 31 else np.int64
        return sp.csr_matrix((vals[mask], cols[mask], indptr.astype(idx_dtype)), shape=(r, c))


class MatchList(_Handle):
    _free = "sg_matchlist_free"

    def dims(self):
        """(n_rows, n_entries, dtype code)"""
        r, m, d = C.c_int64(), C.c_int64(), C.c_int32()
        check(lib().sg_matchlist_dims(self.h, C.byref(r), C.byref(m), C.byref(d)))
        return r.value, m.value, d.value

    def best_master(self, n_cols: int) -> np.ndarray:
        """Per column (duplicate) the row (master) with the largest similarity, lowest row among equals;
        -1 where the column has no entry (string_grouper.py:803-807)."""
        out = np.full(max(n_cols, 1), -1, np.int32)
        check(lib().sg_matchlist_best_master(self.ctx.h, self.h, _ptr(out)))
        return out[:n_cols]

    def group_reps(self, centroid: bool) -> np.ndarray:
        """Representative of every string's group: connected components of the list + group_rep
        'first' / 'centroid' (string_grouper.py:851-904)."""
        n = self.dims()[0]
        out = np.zeros(max(n, 1), np.int32)
        check(lib().sg_matchlist_group_reps(self.ctx.h, self.h, 1 if centroid else 0, _ptr(out)))
        return out[:n]

    def to_host(self):
        """(row_ptr int64[n+1], cols int32[m], vals[m])"""
        r, m, d = C.c_int64(), C.c_int64(), C.c_int32()
        check(lib().sg_matchlist_dims(self.h, C.byref(r), C.byref(m), C.byref(d)))
        row_ptr = np.empty(r.value + 1, np.int64)
        cols = np.empty(max(m.value, 1), np.int32)
        vals = np.empty(max(m.value, 1), code_np_dtype(d.value))
        check(lib().sg_matchlist_to_host(self.ctx.h, self.h, _ptr(row_ptr), _ptr(cols), _ptr(vals)))
        return row_ptr, cols[:m.value], vals[:m.value]


class Context:
    """One per (process, GPU).  ``stream``: an integer hipStream_t (e.g. torch's current stream)."""

    def __init__(self, device: int = 0, stream: Optional[int] = None):
        out = C.c_void_p()
        check(lib().sg_ctx_create(int(device), C.c_void_p(stream) if stream else None, C.byref(out)))
        self.h = out
        self.device = device
        # how often a fit of (documents, dtype) has come by without an idf table being installed (vectorizer._finish_fit:
        # the table is uploaded when the same shape comes a second time); guarded by `lock` -- two threads may share a context
        self.lock = threading.Lock()
        self.idf_fits_seen: dict = {}

    def note_idf_fit(self, n_docs: int, dtype_str: str) -> int:
        """Count one fit of (documents, dtype) that found no idf table; returns how many there have been."""
        with self.lock:
            if len(self.idf_fits_seen) > 64:
                self.idf_fits_seen.clear()
            key = (int(n_docs), dtype_str)
            self.idf_fits_seen[key] = self.idf_fits_seen.get(key, 0) + 1
            return self.idf_fits_seen[key]

    def close(self):
        if self.h is not None:
            lib().sg_ctx_destroy(self.h)
            self.h = None

    def sync(self):
        check(lib().sg_ctx_sync(self.h))

    def trim(self):
        check(lib().sg_ctx_trim(self.h))

    # ---- tuning switches: read from the environment once, when the context is created (include/sg_hip.h)
    def set_option(self, name: str, value=None) -> None:
        """Set (or, with None, unset) one SG_* switch of THIS context."""
        check(lib().sg_ctx_set_option(self.h, name.encode(), None if value is None else str(value).encode()))

    def reset_options(self) -> None:
        """Re-read the SG_* variables of the environment (tests)."""
        check(lib().sg_ctx_reset_options(self.h))

    def options(self) -> dict:
        n = lib().sg_ctx_options(self.h, None, 0)
        buf = C.create_string_buffer(max(int(n), 1))
        lib().sg_ctx_options(self.h, buf, len(buf))
        return dict(line.split("=", 1) for line in buf.value.decode().splitlines() if "=" in line)

    def stats(self) -> dict:
        st = SgStats()
        check(lib().sg_ctx_stats(self.h, C.byref(st)))
        d = {f"ms_{KERNEL_NAMES[i]}": float(st.ms[i]) for i in range(SG_K_COUNT)}
        d.update(macs=int(st.macs), spgemm_bytes=int(st.spgemm_bytes), out_nnz=int(st.out_nnz),
                 prune_rows=int(st.prune_rows), prune_postings=int(st.prune_postings),
                 prune_survivors=int(st.prune_survivors), exact_rows=int(st.exact_rows),
                 prune_bytes=int(st.prune_bytes), prune_symmetric=int(st.prune_symmetric), prune_scored=int(st.prune_scored))
        return d

    # ---- strings
    def strings_from_host(self, data: np.ndarray, offsets: np.ndarray) -> Strings:
        data = np.ascontiguousarray(data, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        out = C.c_void_p()
        n = len(offsets) - 1
        check(lib().sg_strings_from_host(self.h, _ptr(data), _ptr(offsets), n, C.byref(out)))
        s = Strings(self, out)
        s.n = n
        return s

    def strings_from_host_symbols(self, symbols: np.ndarray, offsets: np.ndarray, alphabet_size: int) -> Strings:
        """A symbol column: uint16 ranks in the fit's alphabet (0xFFFF = not in it), offsets in symbols."""
        symbols = np.ascontiguousarray(symbols, dtype=np.uint16)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        out = C.c_void_p()
        n = len(offsets) - 1
        check(lib().sg_strings_from_host_symbols(self.h, _ptr(symbols), _ptr(offsets), n, int(alphabet_size), C.byref(out)))
        s = Strings(self, out)
        s.n = n
        return s

    def strings_set_prelowered(self, s: Strings, prelowered: bool):
        check(lib().sg_strings_set_prelowered(s.h, 1 if prelowered else 0))

    def vocab_coding(self, v: "Vocab"):
        """(bits per character, fitted on symbol columns, sorted vocabulary)"""
        b, sy, so = C.c_int32(), C.c_int32(), C.c_int32()
        check(lib().sg_vocab_coding(v.h, C.byref(b), C.byref(sy), C.byref(so)))
        return b.value, bool(sy.value), bool(so.value)

    def strings_from_device(self, d_bytes: int, d_offsets: int, n: int, total_bytes: int, keepalive=None) -> Strings:
        out = C.c_void_p()
        check(lib().sg_strings_from_device(self.h, C.c_void_p(d_bytes), C.c_void_p(d_offsets), n, total_bytes,
                                           C.byref(out)))
        s = Strings(self, out)
        s.n = n
        s.keepalive = keepalive
        return s

    # ---- vectoriser
    def vec_fit(self, sets, params: SgVecParams) -> Vocab:
        arr = (C.c_void_p * len(sets))(*[s.h for s in sets])
        out = C.c_void_p()
        check(lib().sg_vec_fit(self.h, arr, len(sets), C.byref(params), C.byref(out)))
        return Vocab(self, out)

    def vec_fit_begin(self, sets, params: SgVecParams) -> Vocab:
        """Tokenise + document frequencies of the LOCAL strings; the vocabulary is finished by ``vec_fit_end``."""
        arr = (C.c_void_p * len(sets))(*[s.h for s in sets])
        out = C.c_void_p()
        check(lib().sg_vec_fit_begin(self.h, arr, len(sets), C.byref(params), C.byref(out)))
        return Vocab(self, out)

    def vocab_df_table(self, v: Vocab):
        """(device pointer, entries, shareable) of the dense int32 document-frequency table."""
        p, n, ok = C.c_void_p(), C.c_int64(), C.c_int32()
        check(lib().sg_vocab_df_table(v.h, C.byref(p), C.byref(n), C.byref(ok)))
        return p.value, n.value, bool(ok.value)

    def vec_fit_end(self, v: Vocab, n_docs_total: int = 0):
        check(lib().sg_vec_fit_end(self.h, v.h, int(n_docs_total)))

    def vocab_size(self, v: Vocab):
        a, b = C.c_int64(), C.c_int64()
        check(lib().sg_vocab_size(v.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def vocab_to_host(self, v: Vocab):
        n_terms, _ = self.vocab_size(v)
        keys = np.empty(max(n_terms, 1), np.uint64)
        df = np.empty(max(n_terms, 1), np.int64)
        check(lib().sg_vocab_to_host(self.h, v.h, _ptr(keys), _ptr(df)))
        return keys[:n_terms], df[:n_terms]

    def vocab_byte_alphabet(self, v: Vocab) -> np.ndarray:
        """byte value of every character code of a vocabulary fitted on byte columns"""
        out = np.zeros(128, np.uint8)
        n = C.c_int32()
        check(lib().sg_vocab_byte_alphabet(v.h, _ptr(out), C.byref(n)))
        return out[:n.value]

    def vocab_set_idf(self, v: Vocab, idf: np.ndarray):
        idf = np.ascontiguousarray(idf)
        check(lib().sg_vocab_set_idf(self.h, v.h, _ptr(idf), np_dtype_code(idf.dtype)))

    def put_idf_table(self, n_docs: int, table: np.ndarray) -> None:
        """idf as a function of the document count, table[df] for df = 0 .. n_docs, made by the caller's numpy; the context
        keeps it for the fits over n_docs documents that follow (sg_ctx_put_idf_table)."""
        table = np.ascontiguousarray(table)
        assert table.shape == (n_docs + 1,)
        check(lib().sg_ctx_put_idf_table(self.h, int(n_docs), np_dtype_code(table.dtype), _ptr(table)))

    def vocab_apply_idf_table(self, v: Vocab) -> bool:
        """idf[column] = table[df[column]] on the device; False: no table for this vocabulary's size and dtype yet."""
        ok = C.c_int32()
        check(lib().sg_vocab_apply_idf_table(self.h, v.h, C.byref(ok)))
        return bool(ok.value)

    def vec_transform(self, v: Vocab, s: Strings) -> Csr:
        out = C.c_void_p()
        check(lib().sg_vec_transform(self.h, v.h, s.h, C.byref(out)))
        return Csr(self, out)

    # ---- CSR
    def csr_from_scipy(self, m) -> Csr:
        import scipy.sparse as sp
        m = sp.csr_matrix(m)
        if not m.has_sorted_indices:
            m = m.sorted_indices()
        indptr = np.ascontiguousarray(m.indptr, dtype=np.int64)
        indices = np.ascontiguousarray(m.indices, dtype=np.int32)
        data = np.ascontiguousarray(m.data)
        out = C.c_void_p()
        check(lib().sg_csr_from_host(self.h, m.shape[0], m.shape[1], _ptr(indptr), _ptr(indices), _ptr(data),
                                     np_dtype_code(data.dtype), C.byref(out)))
        return Csr(self, out)

    def csr_from_device(self, n_rows, n_cols, nnz, d_indptr: int, d_indices: int, d_data: int, dtype,
                        keepalive=None) -> Csr:
        out = C.c_void_p()
        check(lib().sg_csr_from_device(self.h, n_rows, n_cols, nnz, C.c_void_p(d_indptr), C.c_void_p(d_indices),
                                       C.c_void_p(d_data), np_dtype_code(dtype), C.byref(out)))
        m = Csr(self, out)
        m.keepalive = keepalive
        return m

    # ---- multiply
    def postings_build(self, B: Csr, tile_cols: int = 0, permute: bool = True) -> Postings:
        """The inverted index of B (sg_postings_build_flags).  By default it is built over the library's fixed row
        permutation -- the ranges of the multi-GPU self-join form are then ranges of POSITIONS (``postings_permutation``,
        ``postings_rows``) -- ``permute=False`` builds it in row order (SG_POSTINGS_NO_PERMUTATION; tests).  All ranks of a
        job must build their index the same way: ``distributed.sharded_selfjoin_topn`` checks it."""
        out = C.c_void_p()
        check(lib().sg_postings_build_flags(self.h, B.h, int(tile_cols), 0 if permute else 1, C.byref(out)))
        return Postings(self, out)

    def spgemm_topn(self, A: Csr, Bt: Postings, top_n: int, threshold: float, sort: bool = True) -> TopN:
        out = C.c_void_p()
        check(lib().sg_spgemm_topn(self.h, A.h, Bt.h, int(top_n), float(threshold), 1 if sort else 0, C.byref(out)))
        return TopN(self, out)

    def matchlist_build(self, res: TopN, fix_diagonal: bool, symmetrize: bool, sort_by_column: bool = False) -> MatchList:
        out = C.c_void_p()
        check(lib().sg_matchlist_build(self.h, res.h, 1 if fix_diagonal else 0, 1 if symmetrize else 0,
                                       1 if sort_by_column else 0, C.byref(out)))
        return MatchList(self, out)

    def rowwise_dot(self, A: Csr, B: Csr) -> np.ndarray:
        """sum_k A[i, k] * B[i, k] per row, in scipy/numpy's arithmetic (string_grouper.py:433-440)."""
        r, _, _, d = A.dims()
        out = np.zeros(max(r, 1), code_np_dtype(d))
        check(lib().sg_csr_rowwise_dot(self.h, A.h, B.h, _ptr(out)))
        return out[:r]

    def selfjoin_range(self, A: Csr, Bt: Postings, top_n: int, threshold: float, row_lo: int, row_hi: int, row_step: int = 1):
        """The rows [row_lo, row_hi) -- with ``row_step`` s > 1: row_hi - 1, row_hi - 1 - s, ... >= row_lo -- of the
        self-join form (include/sg_hip.h: sg_selfjoin_range).  Returns
        (TopN over all rows, device pointer of the mirrored pairs, number of pairs, int32 words per pair), or None when
        the form does not apply to this input."""
        out, pairs = C.c_void_p(), C.c_void_p()
        n_pairs, words, ok = C.c_int64(), C.c_int32(), C.c_int32()
        check(lib().sg_selfjoin_range(self.h, A.h, Bt.h, int(top_n), float(threshold), int(row_lo), int(row_hi),
                                      C.byref(out), C.byref(pairs), C.byref(n_pairs), C.byref(words), C.byref(ok), int(row_step)))
        if not ok.value:
            return None
        return TopN(self, out), pairs.value, n_pairs.value, words.value

    def selfjoin_merge(self, res: TopN, Bt: Optional[Postings], d_pairs: int, n_pairs: int, pair_words: int, row_lo: int,
                       row_hi: int, row_step: int = 1) -> None:
        check(lib().sg_selfjoin_merge(self.h, res.h, Bt.h if Bt is not None else None, C.c_void_p(d_pairs), int(n_pairs),
                                      int(pair_words), int(row_lo), int(row_hi), int(row_step)))

    def postings_permutation(self, Bt: Postings):
        """(device pointer of orig_of, of pos_of), or (0, 0) when the index is in row order."""
        a, b = C.c_void_p(), C.c_void_p()
        check(lib().sg_postings_permutation(Bt.h, C.byref(a), C.byref(b)))
        return a.value or 0, b.value or 0

    def postings_rows(self, Bt: Postings):
        """(rows of the index, rows of the matrix it was built from, device pointer of the row -> group table or 0):
        the first two differ when identical rows were grouped (include/sg_hip.h: sg_postings_rows)."""
        a, b, g = C.c_int64(), C.c_int64(), C.c_void_p()
        check(lib().sg_postings_rows(Bt.h, C.byref(a), C.byref(b), C.byref(g)))
        return a.value, b.value, g.value or 0

    def postings_bytes(self, Bt: Postings) -> int:
        """Bytes of the index the pruned multiply reads while it runs (include/sg_hip.h: sg_postings_bytes)."""
        b = C.c_int64()
        check(lib().sg_postings_bytes(Bt.h, C.byref(b)))
        return int(b.value)

    def topn_expand_groups(self, Bt: Postings, groups: "TopN", d_rows: int = 0, n_rows: int = 0) -> "TopN":
        """The rows ``d_rows`` (device int32 row numbers; 0 = all rows) of the result over the caller's rows, from a
        result with one row per group of the index (sg_topn_expand_groups)."""
        out = C.c_void_p()
        check(lib().sg_topn_expand_groups(self.h, Bt.h, groups.h, C.c_void_p(d_rows) if d_rows else None, int(n_rows), C.byref(out)))
        return TopN(self, out)

    def topn_expand_range(self, Bt: Postings, groups: "TopN", pos_lo: int, pos_hi: int, pos_step: int = 1):
        """(result rows, device pointer of their int32 row numbers -- ``device_free`` it --, how many) for the rows of the
        groups at the positions [pos_lo, pos_hi) of the index (sg_topn_expand_range)."""
        out, rows, n = C.c_void_p(), C.c_void_p(), C.c_int64()
        check(lib().sg_topn_expand_range(self.h, Bt.h, groups.h, int(pos_lo), int(pos_hi), C.byref(out), C.byref(rows), C.byref(n),
                                         int(pos_step)))
        return TopN(self, out), rows.value or 0, n.value

    def device_free(self, d_ptr: int) -> None:
        if d_ptr:
            check(lib().sg_device_free(self.h, C.c_void_p(d_ptr)))

    def row_costs(self, A: Csr, Bt: Postings) -> np.ndarray:
        out = np.zeros(max(A.dims()[0], 1), np.int64)
        check(lib().sg_row_costs(self.h, A.h, Bt.h, _ptr(out)))
        return out[:A.dims()[0]]

    def topn_from_host(self, cols: np.ndarray, vals: np.ndarray, counts: np.ndarray, n_cols: int) -> TopN:
        n_rows, stride = cols.shape
        cols = np.ascontiguousarray(cols, dtype=np.int32)
        vals = np.ascontiguousarray(vals)
        counts = np.ascontiguousarray(counts, dtype=np.int32)
        out = C.c_void_p()
        check(lib().sg_topn_from_host(self.h, n_rows, n_cols, stride, np_dtype_code(vals.dtype), _ptr(cols),
                                      _ptr(vals), _ptr(counts), C.byref(out)))
        return TopN(self, out)

    def topn_from_device(self, n_rows: int, stride: int, n_cols: int, dtype, d_cols: int, d_vals: int, d_counts: int) -> TopN:
        """A fixed-stride result from device pointers (copied on the context's stream; the caller keeps its buffers
        alive until the stream has passed the copy, e.g. ``ctx.sync()``)."""
        out = C.c_void_p()
        check(lib().sg_topn_from_device(self.h, int(n_rows), int(n_cols), int(stride), np_dtype_code(np.dtype(dtype)),
                                        C.c_void_p(d_cols), C.c_void_p(d_vals), C.c_void_p(d_counts), C.byref(out)))
        return TopN(self, out)

    def topn_zip(self, parts, col_offsets, top_n: int) -> TopN:
        arr = (C.c_void_p * len(parts))(*[p.h for p in parts])
        offs = np.ascontiguousarray(col_offsets, dtype=np.int64)
        out = C.c_void_p()
        check(lib().sg_topn_zip(self.h, arr, _ptr(offs), len(parts), int(top_n), C.byref(out)))
        return TopN(self, out)


_default_ctx = {}


def default_context(device: Optional[int] = None) -> Context:
    """Process-wide context per device (device defaults to LOCAL_RANK, else 0)."""
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0"))
    ctx = _default_ctx.get(device)
    if ctx is None or ctx.h is None:
        ctx = Context(device)
        _default_ctx[device] = ctx
    return ctx
