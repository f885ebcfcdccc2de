"""Debug aid: K7/K8 against numpy on the host copy of the same match list."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pandas as pd
import scipy.sparse as sp
from string_grouper_amd import _native as N
import string_grouper_amd.engine as E
from string_grouper_amd.synth import synth_names

ctx = N.default_context(0)
eng = E.HipEngine(ctx)
master = list(synth_names(8000, 31)); master = master + master[:500]
dupes = list(synth_names(5000, seed=32, perturb_of=np.asarray(master, dtype=object), perturb_frac=0.6))
A, B, vec = eng.tfidf(pd.Series(master), pd.Series(dupes), 3, r'[,-./]|\s', True, True, np.float32)
rows, cols, vals, tmax, dml = eng.match_list(A, B, 20, 0.8, False, keep_on_device=True)
bm = dml.best_master()
order = np.lexsort((rows, -vals.astype(np.float64), cols))
cs = cols[order]; first = np.ones(len(order), bool); first[1:] = cs[1:] != cs[:-1]
want = np.full(len(dupes), -1, np.int64); want[cs[first]] = rows[order][first]
bad = np.flatnonzero(bm != want)
print("best_master mismatches", len(bad), "of", len(dupes), "entries", len(rows))
for c in bad[:5]:
    sel = cols == c
    print(" col", c, "dev", bm[c], "want", want[c], "entries", list(zip(rows[sel].tolist(), vals[sel].tolist())))
# group reps
names = list(synth_names(6000, 21))
names += [names[3]] * 40 + [names[5] + " INC"] * 25
A, _, _ = eng.tfidf(pd.Series(names), None, 3, r'[,-./]|\s', True, True, np.float32)
rows, cols, vals, tmax, dml = eng.match_list(A, A, 20, 0.8, True, keep_on_device=True)
n = len(names)
for centroid in (False, True):
    rep = dml.group_reps(centroid)
    g = sp.csr_matrix((np.ones(len(rows)), (rows, cols)), shape=(n, n))
    from scipy.sparse.csgraph import connected_components
    _, labels = connected_components(g, directed=True)
    v64 = vals.astype(np.float64)
    g.data = v64
    weight = np.asarray(g.sum(axis=1)).squeeze(axis=1)
    # numpy's reduce: first element + pairwise sum of the rest
    order = np.lexsort((np.arange(n), -weight, labels)) if centroid else np.lexsort((np.arange(n), labels))
    ls = labels[order]; head = np.ones(n, bool); head[1:] = ls[1:] != ls[:-1]
    rol = np.empty(labels.max() + 1, np.int64); rol[ls[head]] = order[head]
    want = rol[labels]
    bad = np.flatnonzero(rep != want)
    print("group_reps centroid", centroid, "mismatches", len(bad))
    for i in bad[:3]:
        members = np.flatnonzero(labels == labels[i])
        print("  row", i, "dev", rep[i], "want", want[i], "group size", len(members), "weights dev/want", weight[rep[i]].hex(), weight[want[i]].hex())

# through StringGrouper
import string_grouper_amd as sga
E.set_engine(eng)
m, d = pd.Series(master), pd.Series(dupes)
for kw in (dict(min_similarity=0.8), dict(min_similarity=0.6, max_n_matches=5)):
    sg = sga.StringGrouper(m, d, tfidf_matrix_dtype=np.float32, **kw).fit()
    dml = sg.__dict__.get('_device_matches')
    ml = sg._matches_list
    print("kw", kw, "device list?", dml is not None, "entries", len(ml), "n_cols", dml.n_cols if dml else None)
    dml = sg.__dict__.get('_device_matches')
    print("  still there after reading _matches_list?", dml is not None)
    bm = dml.best_master()
    ms, ds, sim = ml.master_side.to_numpy(), ml.dupe_side.to_numpy(), ml.similarity.to_numpy()
    order = np.lexsort((ms, -sim, ds)); cs = ds[order]; first = np.ones(len(order), bool); first[1:] = cs[1:] != cs[:-1]
    want = np.full(len(dupes), -1, np.int64); want[cs[first]] = ms[order][first]
    bad = np.flatnonzero(bm != want)
    print("  mismatches", len(bad))
    for c in bad[:4]:
        sel = ds == c
        print("   col", c, "dev", bm[c], "want", want[c], list(zip(ms[sel].tolist(), sim[sel].tolist())))
