"""CPU model: would INDEX-side pruning pay in the pruned multiply?  Today only the query side is pruned: row i streams the
posting lists of its prefix P_i (all terms but a suffix S_i of its most frequent ones with ||a_S|| <= beta).  With one
global term order (list length descending) the rarest term two rows share lies in BOTH prefixes -- so a posting (j, k)
is only needed in the filter index if k is outside T_j, the gamma-suffix of row j (its most frequent terms with
||b_T|| <= gamma); the survivor test then pays  ||a restricted to its frequent prefix terms|| * gamma  more.
For a sample of rows: postings streamed and survivors, today vs gamma in a few values.

    python scripts/k4p_index_prune_model.py [rows=663000] [sample=400]
"""
import sys
import time

import numpy as np
from sklearn.feature_extraction.text import TfidfVectorizer

sys.path.insert(0, ".")
from string_grouper_amd.synth import synth_names  # noqa: E402


def suffix_mask(m, df_all, budget, freq_min):
    """per non-zero: is it in its row's suffix (most frequent terms while the sum of squares stays <= budget)?"""
    n = m.shape[0]
    rows = np.repeat(np.arange(n), np.diff(m.indptr))
    df = df_all[m.indices]
    pos = np.arange(m.nnz) - m.indptr[rows]
    order = np.lexsort((pos, -df, rows))
    w = (m.data.astype(np.float64) ** 2) * 1.00001
    cs = np.cumsum(w[order])
    row_o = rows[order]
    first = np.r_[0, np.flatnonzero(np.diff(row_o)) + 1]
    base = np.zeros(m.nnz)
    base[first] = np.r_[0.0, cs[first[1:] - 1]]
    base = np.maximum.accumulate(base)
    cum = cs - base
    ins = np.zeros(m.nnz, bool)
    ins[order] = (cum <= budget) & (df[order] >= freq_min)
    return ins


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 663000
    n_sample = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    thr, delta = 0.8, 0.05
    t0 = time.time()
    names = synth_names(n, 1234)
    m = TfidfVectorizer(analyzer="char", ngram_range=(3, 3), lowercase=True, dtype=np.float32).fit_transform(names).tocsr()
    m.sort_indices()
    mt = m.T.tocsr()
    mt.sort_indices()
    df_all = np.diff(mt.indptr)
    freq_min = max(1, int(0.0045 * n))
    beta = thr - delta
    in_s = suffix_mask(m, df_all, beta * beta * (1 - 1e-6), freq_min)
    frequent = df_all >= freq_min
    # f_j: norm of the frequent part of every row
    rows_all = np.repeat(np.arange(n), np.diff(m.indptr))
    f2 = np.bincount(rows_all, weights=(m.data.astype(np.float64) ** 2) * frequent[m.indices], minlength=n)
    f = np.sqrt(f2)
    print(f"# {m.shape}, nnz {m.nnz}, {time.time() - t0:.0f} s", flush=True)
    rng = np.random.default_rng(3)
    sample = np.sort(rng.choice(n, n_sample, replace=False))
    gammas = [0.0, 0.2, 0.3, 0.4, 0.5]
    in_t = {g: (suffix_mask(m, df_all, g * g, freq_min) if g > 0 else np.zeros(m.nnz, bool)) for g in gammas}
    # per posting of the transposed matrix: is (j, k) in T_j?  map through a csr of flags
    import scipy.sparse as sp
    flagT = {}
    for g in gammas:
        fl = sp.csr_matrix((in_t[g].astype(np.int8) + 1, m.indices, m.indptr), shape=m.shape).T.tocsr()
        fl.sort_indices()
        flagT[g] = fl.data == 2          # aligned with mt's postings (same sparsity pattern, sorted)
    tot = {g: dict(post=0, surv=0) for g in gammas}
    matches = 0
    for i in sample:
        lo, hi = m.indptr[i], m.indptr[i + 1]
        k, a = m.indices[lo:hi], m.data[lo:hi].astype(np.float64)
        s_i = in_s[lo:hi]
        if s_i.all():
            continue
        aS = np.sqrt((a[s_i] ** 2).sum())
        aPF = np.sqrt((a[~s_i & frequent[k]] ** 2).sum())       # prefix terms that are frequent: they can lie in some T_j
        full = np.zeros(n)
        for kk, aa in zip(k, a):
            sl = slice(mt.indptr[kk], mt.indptr[kk + 1])
            cols = mt.indices[sl]
            full[cols] += aa * mt.data[sl]
        matches += int(((full > thr) & (np.arange(n) <= i)).sum())
        for g in gammas:
            p = np.zeros(n)
            streamed = 0
            for kk, aa in zip(k[~s_i], a[~s_i]):
                sl = slice(mt.indptr[kk], mt.indptr[kk + 1])
                keep = ~flagT[g][sl]
                cols = mt.indices[sl][keep]
                cols_le = cols <= i
                streamed += int(cols_le.sum())
                p[cols] += aa * mt.data[sl][keep]
            touched = (p > 0) & (np.arange(n) <= i)
            surv = touched & (p > thr - aS * f - aPF * g - 1e-4)
            assert not ((full > thr) & (np.arange(n) <= i) & ~surv).any(), (i, g)      # exactness of the rule
            tot[g]["post"] += streamed
            tot[g]["surv"] += int(surv.sum())
    for g in gammas:
        t = tot[g]
        print(f"gamma {g}: postings streamed {t['post'] / n_sample:.0f} per row ({100 * t['post'] / tot[0.0]['post']:.1f} %), "
              f"pairs scored {t['surv'] / n_sample:.1f} per row ({100 * t['surv'] / max(1, tot[0.0]['surv']):.1f} %); matches {matches / n_sample:.2f} per row")


if __name__ == "__main__":
    main()
