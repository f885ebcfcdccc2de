"""Where the public API's wall-clock goes on the GPU box: string_grouper_amd.match_strings on the headline list under cProfile
(after a warm-up call), the engine's own split beside it.  python scripts/e2e_profile.py [rows=663000] [f32|f64]"""
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
dt = np.float64 if (len(sys.argv) > 2 and sys.argv[2] == "f64") else np.float32
ctx = N.default_context(0)
eng = E.HipEngine(ctx)
E.set_engine(eng)
s = pd.Series(synth_names(n, 1234))
kw = dict(max_n_matches=10, min_similarity=0.8, tfidf_matrix_dtype=dt)
sga.match_strings(s, **kw)
best = None
for _ in range(3):
    t0 = time.perf_counter()
    df = sga.match_strings(s, **kw)
    t = time.perf_counter() - t0
    if best is None or t < best:
        best, split = t, dict(eng.timings)
print(f"{n} names: {best:.4f} s, {len(df)} match rows; engine split {split}")
pr = cProfile.Profile()
pr.enable()
df = sga.match_strings(s, **kw)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)
st.sort_stats("tottime").print_stats(18)
