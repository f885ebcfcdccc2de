"""Which rows of one job of tests/test_parity_gpu.py::test_random_lists_with_repeats_equal_the_port differ from the port?
python scripts/job_diff_probe.py <job> [SG_HIP_LIB is honoured]"""
import os
import sys

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import test_parity_gpu as T  # noqa: E402
from oracle import port as P  # noqa: E402
from string_grouper_amd import _native as N  # noqa: E402

want = int(sys.argv[1])
ctx = N.Context()
rng = np.random.default_rng(2024)
for job in range(want + 1):
    n = int(rng.choice([6000, 9000, 14000, 30000, 70000]))
    names = list(T._names(n, seed=100 + job))
    for _ in range(int(rng.integers(0, 6))):
        hub = names[int(rng.integers(0, n))]
        size = int(rng.choice([3, 20, 64, 65, 200, 1500]))
        for k, at in enumerate(rng.choice(n, min(size, n // 4), replace=False)):
            names[at] = hub if k % 4 else hub + " " + "XYZW"[k % 3]
    share = float(rng.choice([0.0, 0.02, 0.05, 0.3]))
    for at in rng.choice(n - 1, int(share * n), replace=False):
        names[at + 1] = names[at]
    if rng.random() < 0.5:
        names = sorted(names)
    dtype = np.float32 if rng.random() < 0.6 else np.float64
    top_n = int(rng.choice([1, 2, 10, 10, 20, 63, 64, 65, 100, 128]))
    thr = float(rng.choice([0.5, 0.6, 0.75, 0.8, 0.8, 0.9, 0.95]))
    sym = str(rng.choice(["", "0", "1"]))
    lo = int(rng.integers(0, n // 2))
print(f"job {want}: n={n} top_n={top_n} thr={thr} {dtype.__name__} SG_SYM={sym!r} repeats={share}", flush=True)
A = T._tfidf(names, dtype)
dA = ctx.csr_from_scipy(A)
if sym:
    ctx.set_option("SG_SYM", sym)
for opt in sys.argv[2:]:
    k, v = opt.split("=")
    ctx.set_option(k, v)
post = ctx.postings_build(dA)
res = ctx.spgemm_topn(dA, post, top_n, thr, True)
got = res.to_scipy()
ref = P.sp_matmul_topn_port(A, A.T, top_n, thr, True, 16)
bad = [i for i in range(n) if got.indptr[i + 1] - got.indptr[i] != ref.indptr[i + 1] - ref.indptr[i]
       or not np.array_equal(got.indices[got.indptr[i]:got.indptr[i + 1]], ref.indices[ref.indptr[i]:ref.indptr[i + 1]])]
print("rows that differ:", len(bad), bad[:20])
for i in bad[:6]:
    print(f"  row {i} {names[i]!r} nnz {A.indptr[i + 1] - A.indptr[i]} values {A.data[A.indptr[i]:A.indptr[i + 1]].round(4).tolist()}")
    print("     got", got.indices[got.indptr[i]:got.indptr[i + 1]].tolist(), got.data[got.indptr[i]:got.indptr[i + 1]].tolist(),
          " want", ref.indices[ref.indptr[i]:ref.indptr[i + 1]].tolist(), ref.data[ref.indptr[i]:ref.indptr[i + 1]].tolist())
    same = [j for j in range(n) if names[j] == names[i]]
    print("     rows with the same name:", same[:10], len(same))
print("stats", {k: v for k, v in ctx.stats().items() if k.startswith("prune")})
