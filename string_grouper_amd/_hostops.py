"""Host-side helpers of the public API's frames (string_grouper_amd/csrc/sg_hostops.c -> libsg_host.so): the two object
gathers of ``get_matches`` and the string column's UTF-8 bytes, each on a few host threads.  Optional: without the library
(or on an input it does not take) every caller falls back to the numpy / pyarrow code it replaces -- same objects, same
bytes.  Loaded with ``ctypes.PyDLL``: the calling thread keeps the GIL while the helper's threads read (and, in the gather,
count references of) Python objects."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None
_tried = False


def _load():
    global _lib, _tried
    if _tried:
        return _lib
    _tried = True
    if os.environ.get("SG_HOST_HELPERS", "1") == "0":          # A/B and test hook: the numpy / pyarrow code
        return None
    path = os.path.join(_HERE, "libsg_host.so")
    if not os.path.exists(path):
        return None
    try:
        lib = C.PyDLL(path)
        lib.sg_host_gather_objects.restype = C.c_int
        lib.sg_host_gather_objects.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int]
        lib.sg_host_all_exact_str.restype = C.c_int
        lib.sg_host_all_exact_str.argtypes = [C.c_void_p, C.c_int64, C.c_int]
        lib.sg_host_ascii_lengths.restype = C.c_int
        lib.sg_host_ascii_lengths.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int]
        lib.sg_host_ascii_copy.restype = None
        lib.sg_host_ascii_copy.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int]
        lib.sg_host_expand_rows.restype = None
        lib.sg_host_expand_rows.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int]
        lib.sg_host_widen_i32.restype = None
        lib.sg_host_widen_i32.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int]
        lib.sg_host_widen_f32.restype = None
        lib.sg_host_widen_f32.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int]
        lib.sg_host_affine_i64.restype = None
        lib.sg_host_affine_i64.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_int]
        _lib = lib
    except (OSError, AttributeError):
        _lib = None
    return _lib


def _threads() -> int:
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(16, n))


def take_objects(values: np.ndarray, positions: np.ndarray) -> np.ndarray:
    """``values.take(positions)`` for a one-dimensional object array (the same objects, new references)."""
    lib = _load()
    n = len(positions)
    if (lib is None or n < 65536 or values.dtype != object or values.ndim != 1 or not values.flags.c_contiguous):
        return values.take(positions)
    idx = np.ascontiguousarray(positions, dtype=np.int64)
    out = np.empty(n, dtype=object)
    st = lib.sg_host_gather_objects(values.ctypes.data, len(values), idx.ctypes.data, n, out.ctypes.data, _threads())
    if st != 0:
        del out                                     # (an index out of range: numpy raises the error the caller expects)
        return values.take(positions)
    return out


def ascii_column_bytes(values: np.ndarray) -> Optional[Tuple[np.ndarray, np.ndarray]]:
    """(UTF-8 bytes uint8, offsets int64[n + 1]) of an object array whose elements are all ASCII ``str`` -- or None: the
    column holds something else (a non-ASCII string, a non-string) and the general conversion has to look at it."""
    lib = _load()
    n = len(values)
    if lib is None or n < 4096 or values.dtype != object or values.ndim != 1 or not values.flags.c_contiguous:
        return None
    offsets = np.empty(n + 1, dtype=np.int64)
    t = _threads()
    if lib.sg_host_ascii_lengths(values.ctypes.data, n, offsets.ctypes.data, t) != 0:
        return None
    data = np.empty(int(offsets[-1]), dtype=np.uint8)
    lib.sg_host_ascii_copy(values.ctypes.data, n, offsets.ctypes.data, data.ctypes.data, t)
    return data, offsets


def all_exact_str(values: np.ndarray) -> bool:
    """True when every element of the object array is exactly a ``str``; False: something else is in it, or the helper is
    not there (the caller then looks for itself)."""
    lib = _load()
    if lib is None or len(values) < 4096 or values.dtype != object or values.ndim != 1 or not values.flags.c_contiguous:
        return False
    return bool(lib.sg_host_all_exact_str(values.ctypes.data, len(values), _threads()))


_BIG = 262144      # below this numpy's own loop is as fast as starting the threads


def expand_rows(row_ptr: np.ndarray) -> np.ndarray:
    """``np.repeat(np.arange(n), np.diff(row_ptr))`` as int64: the row of every entry of a CSR list."""
    lib = _load()
    n = len(row_ptr) - 1
    m = int(row_ptr[-1]) if n >= 0 and len(row_ptr) else 0
    if (lib is None or m < _BIG or row_ptr.dtype != np.int64 or not row_ptr.flags.c_contiguous or int(row_ptr[0]) != 0):
        return np.repeat(np.arange(max(n, 0), dtype=np.int64), np.diff(row_ptr))
    out = np.empty(m, dtype=np.int64)
    lib.sg_host_expand_rows(row_ptr.ctypes.data, n, out.ctypes.data, _threads())
    return out


def widen(values: np.ndarray, dtype) -> np.ndarray:
    """``values.astype(dtype)`` for int32 -> int64 and float32 -> float64 (what the frames hold); anything else: numpy."""
    lib = _load()
    dtype = np.dtype(dtype)
    n = len(values)
    if lib is None or n < _BIG or values.ndim != 1 or not values.flags.c_contiguous:
        return values.astype(dtype)
    if values.dtype == np.int32 and dtype == np.int64:
        out = np.empty(n, dtype=np.int64)
        lib.sg_host_widen_i32(values.ctypes.data, n, out.ctypes.data, _threads())
        return out
    if values.dtype == np.float32 and dtype == np.float64:
        out = np.empty(n, dtype=np.float64)
        lib.sg_host_widen_f32(values.ctypes.data, n, out.ctypes.data, _threads())
        return out
    return values.astype(dtype)


def affine_i64(values: np.ndarray, start: int, step: int) -> np.ndarray:
    """``start + values.astype(int64) * step`` for an int64 array (always a new array)."""
    lib = _load()
    n = len(values)
    if lib is None or n < _BIG or values.dtype != np.int64 or values.ndim != 1 or not values.flags.c_contiguous:
        v = values.astype(np.int64, copy=True)
        return v if (start == 0 and step == 1) else start + v * step
    out = np.empty(n, dtype=np.int64)
    lib.sg_host_affine_i64(values.ctypes.data, n, int(start), int(step), out.ctypes.data, _threads())
    return out
