"""How many of the columns the stream form records are scored in vain, and what a second filter on an 8-bit copy of the
candidate's row would reject (CPU model: tests/test_prune_model.py's restatement of the kernel's filter; self-join form).
python scripts/k4p_second_filter_model.py [names=20000] [every=10]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import test_prune_model as T  # noqa: E402
from oracle import oracle as O  # noqa: E402
from string_grouper_amd.synth import synth_names  # noqa: E402

f32 = np.float32
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
every = int(sys.argv[2]) if len(sys.argv) > 2 else 10
names = synth_names(n, 1234)
(m,), _, _ = O.tfidf_sklearn(names, [names], dtype=np.float32)
m = m.tocsr()
m.sort_indices()
mt = m.T.tocsr()
mt.sort_indices()
norm_up = np.nextafter(f32(np.sqrt(f32(np.asarray(m.multiply(m).sum(axis=1)).max())) * f32(1.000001)), f32(2))
thr, delta, freq = 0.8, 0.03, 0.005
freq_min = max(1, int(freq * n))
fq, bq, post = T.quantise_right_stream(m, mt, freq_min, norm_up, tile=4096)
C = O.sp_matmul_topn(m, m.T.tocsr(), 100000, thr, sort=True)
rng = np.random.default_rng(5)
t0 = time.time()
records = distinct = true = passed8 = passed16 = rows = 0


def bucket_of(k):      # the kernel's term_hash (sg_spgemm_pruned.hip): a bucket of four slots
    return ((((int(k) & 0xffffff) * 0x9E3779) & 0xffffffff) >> 15) & 124
md = m.data.astype(np.float64)
for i in range(0, n, every):
    lo, hi = m.indptr[i], m.indptr[i + 1]
    rec, _ = T.records_of_row_stream(m.indices[lo:hi], m.data[lo:hi], mt.indptr, mt.indices, bq, fq, post, thr, delta,
                                     norm_up, freq_min, rng, tile=4096)
    if rec is None:
        continue
    rows += 1
    rec = rec[rec <= i]                      # self-join form: the pairs j <= i are row i's
    uni = np.unique(rec)
    want = C.indices[C.indptr[i]:C.indptr[i + 1]]
    records += len(rec)
    distinct += len(uni)
    true += int((want <= i).sum())
    a = dict(zip(m.indices[lo:hi].tolist(), md[lo:hi].tolist()))
    # a 16-BIT entry (what a next round could try: one line per record, half the trips): 5 bits bucket, 5 bits fingerprint
    # (the term's low bits), 6 bits value -- a lookup matches any term of row i in the bucket with the same fingerprint
    by_cell = {}
    for k, v in a.items():
        cell = (bucket_of(k), k & 31)
        by_cell[cell] = max(by_cell.get(cell, 0.0), v)
    for j in uni:                            # upper bound of the score from row j's values rounded UP to 8 bits
        jl, jh = m.indptr[j], m.indptr[j + 1]
        ub = sum(a[k] * (np.ceil(v / float(norm_up) * 255) / 255 * float(norm_up))
                 for k, v in zip(m.indices[jl:jh].tolist(), md[jl:jh].tolist()) if k in a)
        passed8 += ub > thr - 1e-5
        ub16 = sum(by_cell.get((bucket_of(k), k & 31), 0.0) * (np.ceil(v / float(norm_up) * 63) / 63 * float(norm_up))
                   for k, v in zip(m.indices[jl:jh].tolist(), md[jl:jh].tolist()))
        passed16 += ub16 > thr - 1e-5
print(f"{rows} rows of {n} names (threshold {thr}, delta {delta}): {records} records, {distinct} distinct columns, "
      f"{true} pairs above the threshold; an 8-bit copy of the candidate's row passes {passed8} of the {distinct} "
      f"({100.0 * passed8 / max(distinct, 1):.0f} %); a 16-bit entry (bucket 5 + fingerprint 5 + value 6 bits) would pass {passed16} "
      f"({100.0 * passed16 / max(distinct, 1):.1f} %)   [{time.time() - t0:.0f} s]")
