"""Where the wall-clock of the other two public functions goes on the GPU box: group_similar_strings (K8 on the device list)
and match_most_similar (K7) on the headline list under cProfile, after a warm-up call.
python scripts/e2e_profile_groups.py [rows=663000]"""
import cProfile
import pstats
import sys
import time

import numpy as np
import pandas as pd

sys.path.insert(0, ".")
import string_grouper_amd as sga  # noqa: E402
import string_grouper_amd.engine as E  # noqa: E402
from string_grouper_amd import _native as N  # noqa: E402
from string_grouper_amd.synth import synth_names  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 663000
ctx = N.default_context(0)
E.set_engine(E.HipEngine(ctx))
names = synth_names(n, 1234)
s = pd.Series(names)
dupes = pd.Series(synth_names(n // 4, 8, perturb_of=names, perturb_frac=0.6))
jobs = (("group_similar_strings", lambda: sga.group_similar_strings(s, min_similarity=0.8, tfidf_matrix_dtype=np.float32)),
        ("match_most_similar", lambda: sga.match_most_similar(s, dupes, min_similarity=0.8, tfidf_matrix_dtype=np.float32)))
for label, fn in jobs:
    fn()
    best = min((lambda t0: (fn(), time.perf_counter() - t0)[1])(time.perf_counter()) for _ in range(3))
    print(f"== {label}: {n} names: {best:.4f} s")
    pr = cProfile.Profile()
    pr.enable()
    fn()
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(14)
