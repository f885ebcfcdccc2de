"""Round 6: the exact kernel in the self-join form (thresholds below the pruned kernel's envelope) -- kernel and pass times
per tile size of the index.   python scripts/exact_sym_probe.py [rows=200000] [thr=0.4]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from string_grouper_amd import _native as N  # noqa: E402
from string_grouper_amd.synth import synth_names  # noqa: E402
from string_grouper_amd.vectorizer import HipTfidfVectorizer  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
thrs = [float(x) for x in sys.argv[2:]] or [0.4]
ctx = N.default_context(0)
vec = HipTfidfVectorizer(dtype=np.float32, ctx=ctx)
p = vec.prepare(synth_names(n, 77))
vec.fit_prepared([p])
A = vec.transform_prepared(p)
for label, opts in (("index as built for the pruned multiply (4096-column tiles)", {}),
                    ("exact kernel, self-join form, whatever the threshold", {"SG_PRUNE_MIN_THRESHOLD": "0.99"}),
                    ("SG_PRUNE_TILE=11 (2048)", {"SG_PRUNE_TILE": "11"}),
                    ("one-sided, pruned index", {"SG_EXACT_SYM": "0"}),
                    ("one-sided, SG_PRUNE=0 index (2048)", {"SG_PRUNE": "0"})):
    for k, v in opts.items():
        ctx.set_option(k, v)
    post = ctx.postings_build(A)
    for thr in thrs:
        best = None
        for _ in range(3):
            r = ctx.spgemm_topn(A, post, 10, thr, True)
            ctx.sync()
            st = ctx.stats()
            r.free()
            if best is None or st["ms_spgemm_topn"] < best["ms_spgemm_topn"]:
                best = st
        print(f"{label:60s} thr {thr}: multiply {best['ms_spgemm_topn']:7.2f} ms, kernel {best['ms_spgemm_kernel']:7.2f} ms, "
              f"sym={best['prune_symmetric']} matches {best['out_nnz']}", flush=True)
    post.free()
    ctx.reset_options()
