"""CPU model of FOLDED accumulator tiles for the pruned multiply (K4p): a visit covers a super-tile of F x 4096
columns whose partial sums are accumulated in ONE 4096-entry tile (column c of the super-tile -> accumulator c mod 4096).
Sums of several columns in one accumulator are still upper bounds of each of them, so the filter stays exact as long as
every posting whose accumulator reaches ITS column's survivor threshold records the column (`new >= tq_j`, not the
`old < tq_j <= new` crossing test of the unfolded form -- with two columns in one accumulator the crossing may be made by
the other column's posting).  Price: false positives (two unrelated columns adding up) and duplicate records (every
posting of a column after the first that reaches the threshold fires); both are scored exactly, so neither changes the
result.  This script measures, for a sample of left rows at the headline size:
  * visits per row, postings per visit, slot utilisation and overflow visits for (F, slots per lane);
  * records / distinct columns recorded / true survivors of the unfolded rule.

    python scripts/k4f_fold_model.py [rows=663000] [sample=1500] [threshold=0.8] [delta=0.05]
"""
import sys
import time

import numpy as np
from sklearn.feature_extraction.text import TfidfVectorizer

sys.path.insert(0, ".")
from string_grouper_amd.synth import synth_names  # noqa: E402

f32 = np.float32
TILE = 4096


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 663000
    n_sample = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
    thr = float(sys.argv[3]) if len(sys.argv) > 3 else 0.8
    delta = float(sys.argv[4]) if len(sys.argv) > 4 else 0.05
    t0 = time.time()
    names = synth_names(n, 1234)
    vec = TfidfVectorizer(analyzer="char", ngram_range=(3, 3), lowercase=True, dtype=np.float32)
    m = vec.fit_transform(names).tocsr()
    m.sort_indices()
    # position space: the index is built over the permutation pos_of[j] = j * M mod n
    mult = int(0.6180339887498949 * n) | 1
    while np.gcd(mult, n) != 1:
        mult += 2
    pos_of = (np.arange(n, dtype=np.int64) * mult) % n
    orig_of = np.empty(n, np.int64)
    orig_of[pos_of] = np.arange(n)
    m = m[orig_of]            # rows in position order
    mt = m.T.tocsr()
    mt.sort_indices()
    print(f"# tf-idf {m.shape}, nnz {m.nnz}, {time.time() - t0:.1f} s", flush=True)
    df_all = np.diff(mt.indptr)
    freq_min = max(1, int(0.0045 * n))
    norm_up = float(np.sqrt(np.asarray(m.multiply(m).sum(axis=1)).max())) * 1.000001
    inv = 1.0 / norm_up
    frequent = df_all[m.indices] >= freq_min
    f2 = np.zeros(n)
    np.add.at(f2, np.repeat(np.arange(n), np.diff(m.indptr))[frequent], m.data[frequent].astype(np.float64) ** 2)
    fq = np.minimum(255, np.ceil(np.sqrt(f2) * inv * 255.0 * 1.000002)).astype(np.int64)
    beta = thr - delta
    budget = (beta / norm_up) ** 2 * (1.0 - 1e-6)
    rng = np.random.default_rng(7)
    rows = np.sort(rng.choice(n, n_sample, replace=False))

    CONFIGS = [(1, 4, 11), (2, 8, 10), (3, 8, 9), (4, 8, 9), (4, 12, 9), (3, 12, 9), (6, 16, 8), (8, 16, 8)]   # (F, slots per lane, bq bits)
    res = {c: dict(visits=0, post=0, slots=0, used=0, over=0, rec=0, distinct=0, true=0, post_over=0) for c in CONFIGS}
    n_rows = 0
    for i in rows:
        lo, hi = m.indptr[i], m.indptr[i + 1]
        k = m.indices[lo:hi]
        a = m.data[lo:hi].astype(np.float64)
        nnz = len(k)
        if nnz == 0 or nnz > 64:
            continue
        df = df_all[k].astype(np.int64)
        w = a * a * 1.00001
        order = np.lexsort((np.arange(nnz), -df))
        cum = np.empty(nnz)
        cum[order] = np.cumsum(w[order])
        in_s = (cum <= budget) & (df >= freq_min)
        in_p = ~in_s
        npp = int(in_p.sum())
        if npp == 0:
            continue
        n_rows += 1
        bs2 = cum[in_s].max() if in_s.any() else 0.0
        b_s = np.sqrt(bs2) * 1.000002
        t0_ = (thr - 1e-5) * 32768.0 - 2.0
        c1 = b_s * norm_up * (32768.0 / 255.0) * 1.000002
        T0 = int(np.floor(t0_ * 256.0)) - 256 * npp
        C1 = int(c1 * 256.0) + 1
        dfp = df[in_p]
        dsum = float(dfp.sum())
        G = 1 + np.floor((64 - npp) * 0.999 * (dfp / dsum)).astype(np.int64)
        terms = k[in_p]
        ap = a[in_p]
        # all postings of the prefix terms with column <= i's tile end (self-join form)
        cols_l, x_l, term_l = [], [], []
        for q, term in enumerate(terms):
            c = mt.indices[mt.indptr[term]:mt.indptr[term + 1]]
            b = mt.data[mt.indptr[term]:mt.indptr[term + 1]].astype(np.float64)
            cols_l.append(c)
            x_l.append((ap[q], b))
            term_l.append(np.full(len(c), q))
        cols = np.concatenate(cols_l)
        tq_col = (T0 - C1 * fq) >> 8          # per column
        for cfg in CONFIGS:
            F, S, bits = cfg
            r = res[cfg]
            bq_max = (1 << bits) - 1
            sup = F * TILE
            t_end = i // sup + 1
            lim = t_end * sup
            xs = []
            for (aq, b) in x_l:
                bq = np.minimum(bq_max, np.ceil(b * inv * bq_max * 1.000002))
                xs.append(np.floor(aq * norm_up * 32768.0 / bq_max * 1.000002 * bq))      # upper bound of a*b*2^15 (less <1)
            x = np.concatenate(xs)
            tm = np.concatenate(term_l)
            keep = cols < lim
            c_, x_, t_ = cols[keep], x[keep], tm[keep]
            r["visits"] += t_end
            r["post"] += len(c_)
            r["slots"] += t_end * 64 * S
            # slot use / overflow per (term, supertile)
            cnt = np.zeros((npp, t_end), np.int64)
            np.add.at(cnt, (t_, c_ // sup), 1)
            cap = (S * G)[:, None]
            r["used"] += int(np.minimum(cnt, cap).sum())
            ov = (cnt > cap).any(axis=0)
            r["over"] += int(ov.sum())
            r["post_over"] += int(cnt[:, ov].sum())
            # accumulate: order = by term (any order is a valid execution)
            acc_key = (c_ // sup) * TILE + (c_ % TILE)          # (visit, accumulator)
            o = np.lexsort((t_, acc_key))
            ak, cc, xx = acc_key[o], c_[o], x_[o]
            # running sum inside each accumulator
            starts = np.r_[0, np.nonzero(np.diff(ak))[0] + 1]
            cs = np.cumsum(xx)
            base = np.repeat(cs[starts] - xx[starts], np.diff(np.r_[starts, len(ak)]))
            new = cs - base
            fire = new >= tq_col[cc]
            fire &= cc <= i
            r["rec"] += int(fire.sum())
            r["distinct"] += len(np.unique(cc[fire]))
            # unfolded truth: per-column sums
            o2 = np.argsort(c_, kind="stable")
            c2, x2 = c_[o2], x_[o2]
            st2 = np.r_[0, np.nonzero(np.diff(c2))[0] + 1]
            sums = np.add.reduceat(x2, st2) if len(c2) else np.zeros(0)
            uc = c2[st2] if len(c2) else c2
            r["true"] += int(((sums >= tq_col[uc]) & (uc <= i)).sum())
    print(f"rows sampled {n_rows}")
    print("F slots bq | visits/row post/visit slot-use overflow-visits% (postings in them %) | records distinct true (per row)")
    for cfg in CONFIGS:
        r = res[cfg]
        F, S, bits = cfg
        print(f"{F} {S:2d} {bits:2d} | {r['visits'] / n_rows:6.1f} {r['post'] / r['visits']:7.1f} {100 * r['used'] / r['slots']:5.1f}% "
              f"{100 * r['over'] / r['visits']:5.2f}% ({100 * r['post_over'] / r['post']:4.1f}%) | "
              f"{r['rec'] / n_rows:7.1f} {r['distinct'] / n_rows:7.1f} {r['true'] / n_rows:7.1f}   postings/row {r['post'] / n_rows:.0f}")


if __name__ == "__main__":
    main()
