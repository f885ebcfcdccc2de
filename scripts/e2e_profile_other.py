import cProfile, pstats, sys, time
import numpy as np, pandas as pd
sys.path.insert(0, ".")
import string_grouper_amd as sga
import string_grouper_amd.engine as E
from string_grouper_amd import _native as N
from string_grouper_amd.synth import synth_names
ctx = N.default_context(0)
E.set_engine(E.HipEngine(ctx))
names = synth_names(663000, 1234)
s = pd.Series(names)
d = pd.Series(synth_names(663000, 8, perturb_of=names, perturb_frac=0.6))
dupes = d[:166000].reset_index(drop=True)
jobs = (("compute_pairwise_similarities", lambda: sga.compute_pairwise_similarities(s, d, tfidf_matrix_dtype=np.float32)),
        ("match_strings(master, duplicates)", lambda: sga.match_strings(s, dupes, max_n_matches=10, min_similarity=0.8, tfidf_matrix_dtype=np.float32)))
for label, fn in jobs:
    fn()
    best = min((lambda t0: (fn(), time.perf_counter() - t0)[1])(time.perf_counter()) for _ in range(3))
    print(f"== {label}: {best:.4f} s")
    pr = cProfile.Profile(); pr.enable(); fn(); pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(8)
