"""CPU model of what the pruned multiply (K4p, self-join form) streams at the headline size: for a sample of left rows,
the suffix / prefix split of the kernel (sg_spgemm_pruned.hip) and, per (row, column tile) visit, the postings of the
prefix terms -- who they belong to (rare terms vs frequent terms that did not fit the suffix budget), how full the
4 slots x 64 lanes of a visit are, how many visits are empty.  No GPU; used to decide where the kernel's work can still
be cut ("what bounds it now").

    python scripts/k4p_model_stats.py [rows=663000] [sample=3000] [threshold=0.8] [delta=0.05]
"""
import sys
import time

import numpy as np
import scipy.sparse as sp
from sklearn.feature_extraction.text import TfidfVectorizer

sys.path.insert(0, ".")
from string_grouper_amd.synth import synth_names  # noqa: E402

f32 = np.float32


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 663000
    n_sample = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
    thr = float(sys.argv[3]) if len(sys.argv) > 3 else 0.8
    delta = float(sys.argv[4]) if len(sys.argv) > 4 else 0.05
    tile = 4096
    t0 = time.time()
    names = synth_names(n, 1234)
    vec = TfidfVectorizer(analyzer="char", ngram_range=(3, 3), lowercase=True, dtype=np.float32)
    m = vec.fit_transform(names).tocsr()
    m.sort_indices()
    mt = m.T.tocsr()
    mt.sort_indices()
    print(f"# tf-idf {m.shape}, nnz {m.nnz}, {time.time() - t0:.1f} s", flush=True)
    df_all = np.diff(mt.indptr)
    freq_min = max(1, int(0.0045 * n))
    norm_up = float(np.sqrt(np.asarray(m.multiply(m).sum(axis=1)).max())) * 1.000001
    beta = thr - delta
    budget = (beta / norm_up) ** 2 * (1.0 - 1e-6)
    rng = np.random.default_rng(7)
    rows = np.sort(rng.choice(n, n_sample, replace=False))
    tot = dict(rows=0, visits=0, empty=0, post=0, post_rare=0, post_freq=0, np_=0, nS=0, overflow=0, nnz=0, slots_used=0,
               visits_le64=0, visits_le128=0, post_all=0)
    per_visit = []
    TAUS = (0.4, 0.5, 0.6, 0.75, 0.9, 1.1)
    cls = {tau: [0, 0, 0] for tau in TAUS}
    np_hist = np.zeros(65, np.int64)
    for i in rows:
        lo, hi = m.indptr[i], m.indptr[i + 1]
        k = m.indices[lo:hi]
        a = m.data[lo:hi]
        nnz = len(k)
        if nnz == 0 or nnz > 64:
            continue
        df = df_all[k].astype(np.int64)
        w = (a * a * f32(1.00001)).astype(np.float64)
        # order: list length descending, position ascending; cum inclusive
        order = np.lexsort((np.arange(nnz), -df))
        cum = np.empty(nnz)
        cum[order] = np.cumsum(w[order])
        in_s = (cum <= budget) & (df >= freq_min)
        in_p = ~in_s
        npp = int(in_p.sum())
        if npp == 0:
            continue
        tot["rows"] += 1
        tot["nnz"] += nnz
        tot["np_"] += npp
        tot["nS"] += int(in_s.sum())
        np_hist[npp] += 1
        t_end = (i >> 12) + 1
        # postings of every prefix term per tile (columns j <= tile end of the row's own tile: the kernel walks whole tiles)
        counts = np.zeros((npp, t_end), np.int64)
        dfp = df[in_p]
        for q, term in enumerate(k[in_p]):
            cols = mt.indices[mt.indptr[term]:mt.indptr[term + 1]]
            cols = cols[cols < t_end * tile]
            counts[q] = np.bincount(cols >> 12, minlength=t_end)[:t_end]
        per_tile = counts.sum(axis=0)
        tot["visits"] += t_end
        tot["empty"] += int((per_tile == 0).sum())
        tot["post"] += int(per_tile.sum())
        rare = dfp < freq_min
        tot["post_rare"] += int(counts[rare].sum())
        tot["post_freq"] += int(counts[~rare].sum())
        tot["post_all"] += int(sum(min(df_all[t], 1 << 62) for t in k) * (t_end * tile / n) if False else 0)
        tot["visits_le64"] += int((per_tile <= 64).sum())
        tot["visits_le128"] += int((per_tile <= 128).sum())
        # lanes dealt in proportion to list lengths (kernel: G = 1 + floor((64 - np) * 0.999 * df / dsum))
        dsum = float(dfp.sum())
        G = 1 + np.floor((64 - npp) * 0.999 * (dfp / dsum)).astype(np.int64)
        over = np.maximum(counts - 4 * G[:, None], 0)
        ov_visit = (over > 0).any(axis=0)
        tot["overflow"] += int(ov_visit.sum())
        # the slot-by-slot loop: one posting per lane and round trip, until the longest remainder is through
        rounds = np.ceil(over / G[:, None]).max(axis=0)
        tot["slow_rounds"] = tot.get("slow_rounds", 0) + int(rounds.sum())
        tot["slow_rounds16"] = tot.get("slow_rounds16", 0) + int(np.ceil(over / (4 * G[:, None])).max(axis=0).sum())
        tot["post_in_slow"] = tot.get("post_in_slow", 0) + int(per_tile[ov_visit].sum())
        tot["slots_used"] += int(np.minimum(counts, 4 * G[:, None]).sum())
        # row classifier: expected entries per tile of the fullest term against its lanes' four slots
        load = float(((dfp / (n / tile)) / (4.0 * G)).max())
        for tau in TAUS:
            if load > tau:
                cls[tau][0] += 1
                cls[tau][1] += t_end
                cls[tau][2] += int(ov_visit.sum())
        per_visit.append(per_tile)
    pv = np.concatenate(per_visit)
    r = tot["rows"]
    print(f"rows sampled {r}: nnz/row {tot['nnz'] / r:.1f}, prefix terms/row {tot['np_'] / r:.2f}, suffix terms/row {tot['nS'] / r:.2f}")
    print(f"visits/row {tot['visits'] / r:.1f}; postings/row {tot['post'] / r:.0f} = {tot['post'] / tot['visits']:.1f} per visit "
          f"(rare terms {100 * tot['post_rare'] / tot['post']:.1f} %, frequent terms outside the suffix {100 * tot['post_freq'] / tot['post']:.1f} %)")
    print(f"empty visits {100 * tot['empty'] / tot['visits']:.1f} %; visits with <= 64 postings {100 * tot['visits_le64'] / tot['visits']:.1f} %, "
          f"<= 128: {100 * tot['visits_le128'] / tot['visits']:.1f} %; visits with an overflowing term {100 * tot['overflow'] / tot['visits']:.2f} %")
    print(f"overflow visits: {tot['slow_rounds'] / max(1, tot['overflow']):.1f} slot-by-slot rounds each on average "
          f"({tot['slow_rounds'] / tot['visits']:.2f} per visit over all visits; with 16-byte rounds: "
          f"{tot['slow_rounds16'] / tot['visits']:.2f}); they hold {100 * tot['post_in_slow'] / tot['post']:.1f} % of the postings")
    for tau in TAUS:
        c = cls[tau]
        print(f"rows with fullest-term load > {tau}: {100 * c[0] / r:.1f} % of rows, {100 * c[1] / tot['visits']:.1f} % of visits, "
              f"hold {100 * c[2] / max(1, tot['overflow']):.1f} % of the overflow visits; overflow share inside: {100 * c[2] / max(1, c[1]):.1f} %")
    print("postings per visit, percentiles 10/25/50/75/90/99:", np.percentile(pv, [10, 25, 50, 75, 90, 99]).round(0).tolist())
    print("prefix-term count histogram (np: rows):", {int(x): int(c) for x, c in enumerate(np_hist) if c})
    est_total = tot["post"] / r * n
    print(f"extrapolated postings streamed at {n} rows: {est_total:.3e} (bench.py pruning.postings_streamed: 5.27e9 at 663 k)")


if __name__ == "__main__":
    main()
