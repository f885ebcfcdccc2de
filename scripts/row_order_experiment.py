"""EXPERIMENT (round 6; measured and NOT kept -- profiles/r06_row_order_experiment.log: sorting the rows of a band by
their longest prefix list is SLOWER, 7.67 -> 8.1 ... 11.2 ms, the neighbours then queue on the same accumulators' cache lines and
the same long lists end together; the debug hook it calls, sg_debug_set_row_order -- an optional
table of positions read by the whole-matrix self-join launch instead of `sym_hi - 1 - rr` --, was a local patch of
sg_api.hip / sg_internal.h / sg_spgemm_pruned.hip and is NOT in the library: the script documents the experiment, it does
not run against this tree): does the order in which the self-join pass takes its rows matter?  Rows that share their longest
prefix list, taken at the same time, stream that list together (L2 hits instead of fabric traffic).  The order is computed
on the host here and handed to the library through a debug hook (sg_debug_set_row_order); SG_COLLAPSE=0 so that positions
are rows.   python scripts/row_order_experiment.py [rows=663000]"""
import ctypes as C
import math
import sys

import numpy as np

sys.path.insert(0, ".")
from string_grouper_amd import _native as N  # noqa: E402
from string_grouper_amd.synth import synth_names  # noqa: E402
from string_grouper_amd.vectorizer import HipTfidfVectorizer  # noqa: E402

n_names = int(sys.argv[1]) if len(sys.argv) > 1 else 663000
ctx = N.Context()
ctx.set_option("SG_COLLAPSE", "0")
vec = HipTfidfVectorizer(dtype=np.float32, ctx=ctx)
p = vec.prepare(synth_names(n_names, 1234))
vec.fit_prepared([p])
A = vec.transform_prepared(p)
post = ctx.postings_build(A)
H = A.to_scipy()
n = H.shape[0]
df = np.bincount(H.indices, minlength=H.shape[1])
freq_min = max(1, int(0.005 * n))
d = df[H.indices].astype(np.int64)
comp = np.where(d < freq_min, (d << 32) | H.indices.astype(np.int64), -1)
lens = np.diff(H.indptr)
key = np.full(n, -1, dtype=np.int64)
nz = lens > 0
key[nz] = np.maximum.reduceat(comp, H.indptr[:-1][nz])
top_term = np.where(key >= 0, key & 0xFFFFFFFF, H.shape[1]).astype(np.int64)
M = int(0.6180339887498949 * n) | 1
while math.gcd(M, n) != 1:
    M += 2
pos_of = (np.arange(n, dtype=np.uint64) * np.uint64(M % n) % np.uint64(n)).astype(np.int64)
orig_of = np.empty(n, dtype=np.int64)
orig_of[pos_of] = np.arange(n)
key_by_pos = top_term[orig_of]
lib = N.lib()
lib.sg_debug_set_row_order.restype = C.c_int
lib.sg_debug_set_row_order.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]


def run(order, label):
    if order is None:
        lib.sg_debug_set_row_order(ctx.h, None, 0)
    else:
        o = np.ascontiguousarray(order, dtype=np.uint32)
        assert len(o) == n and len(np.unique(o)) == n
        lib.sg_debug_set_row_order(ctx.h, o.ctypes.data, n)
    ms = []
    res = None
    for _ in range(5):
        if res is not None:
            res.free()
        res = ctx.spgemm_topn(A, post, 10, 0.8, True)
        ctx.sync()
        ms.append(ctx.stats()["ms_spgemm_kernel"])
    out = res.to_scipy()
    res.free()
    print(f"{label:48s} kernel {min(ms):8.3f} ms (median {np.median(ms):8.3f})", flush=True)
    return out


base = run(None, "library order (from the last position down)")
desc = np.arange(n - 1, -1, -1)
same = run(desc, "the same order through the table")
for band in (n, 262144, 65536, 16384):
    order = []
    for hi in range(n, 0, -band):
        lo = max(0, hi - band)
        pos = np.arange(hi - 1, lo - 1, -1)
        order.append(pos[np.argsort(key_by_pos[pos], kind="stable")])
    got = run(np.concatenate(order), f"bands of {band} positions, by longest prefix list")
    assert (got != base).nnz == 0 and np.array_equal(got.indptr, base.indptr), "results differ"
rng = np.random.default_rng(0)
run(rng.permutation(n), "random order")
lib.sg_debug_set_row_order(ctx.h, None, 0)
