"""Round 6: the pruned multiply's two forms over the threshold -- stream form (4096-column tiles folded eight to an
accumulator tile, second filter) against the tile-by-tile form (2048 / 4096 columns, exact per-column accumulators) -- and the
exact kernel in the self-join form.   python scripts/form_sweep.py [rows=200000] [top_n=10]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from string_grouper_amd import _native as N  # noqa: E402
from string_grouper_amd.synth import synth_names  # noqa: E402
from string_grouper_amd.vectorizer import HipTfidfVectorizer  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
top_n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
thrs = [float(x) for x in sys.argv[3:]] or [0.4, 0.45, 0.5, 0.55, 0.6, 0.65, 0.7, 0.75, 0.8, 0.9]
ctx = N.default_context(0)
vec = HipTfidfVectorizer(dtype=np.float32, ctx=ctx)
p = vec.prepare(synth_names(n, 77))
vec.fit_prepared([p])
A = vec.transform_prepared(p)
forms = (("stream form (default)", {"SG_PRUNE_MIN_THRESHOLD": "0.3", "SG_ALT_FORM": "0"}),
         ("tile-by-tile, 2048 columns", {"SG_PRUNE_TILE": "11", "SG_PRUNE_MIN_THRESHOLD": "0.3", "SG_ALT_FORM": "0"}),
         ("tile-by-tile, 4096 columns", {"SG_K4_STREAM": "0", "SG_PRUNE_MIN_THRESHOLD": "0.3", "SG_ALT_FORM": "0"}),
         ("exact kernel, self-join form", {"SG_PRUNE_MIN_THRESHOLD": "0.99", "SG_ALT_FORM": "0"}),
         ("library's choice", {}))
table = {}
for label, opts in forms:
    for k, v in opts.items():
        ctx.set_option(k, v)
    post = ctx.postings_build(A)
    ctx.sync()
    build_ms = ctx.stats()["ms_postings"]
    for thr in thrs:
        best = None
        for _ in range(3):
            r = ctx.spgemm_topn(A, post, top_n, thr, True)
            ctx.sync()
            st = ctx.stats()
            r.free()
            if best is None or st["ms_spgemm_topn"] < best["ms_spgemm_topn"]:
                best = st
        table[(label, thr)] = best
    print(f"{label:32s} index {build_ms:6.2f} ms | " + " ".join(f"{table[(label, t)]['ms_spgemm_topn']:7.2f}" for t in thrs), flush=True)
    if os.environ.get("FORM_SWEEP_KERNEL"):
        print(f"{'   of which the kernel':32s}                 | " + " ".join(f"{table[(label, t)]['ms_spgemm_kernel']:7.2f}" for t in thrs), flush=True)
    post.free()
    ctx.reset_options()
print(f"{'threshold':32s}                 | " + " ".join(f"{t:7.2f}" for t in thrs))
print(f"{'matches':32s}                 | " + " ".join(f"{table[(forms[0][0], t)]['out_nnz']:7d}" for t in thrs))
