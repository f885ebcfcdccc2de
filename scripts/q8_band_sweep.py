"""Round 6: where does the second filter (8-bit rows) stop paying?  sg_postings_build builds the records for rows of up to
40 entries on average (SG_Q8=1 forces them, =0 drops them).  This sweep moves the row length through that bar -- three
names joined and cut at L characters, so a row has about L - 2 entries -- and times the self-join with and without.
python scripts/q8_band_sweep.py [rows=100000] [thr=0.8]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from string_grouper_amd import _native as N  # noqa: E402
from string_grouper_amd.synth import synth_names  # noqa: E402
from string_grouper_amd.vectorizer import HipTfidfVectorizer  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 0.8
ctx = N.default_context(0)
base = synth_names(n * 3, 5)
joined = [" ".join(base[3 * i:3 * i + 3]) for i in range(n)]
print(f"{n} rows, threshold {thr}: multiply ms (kernel ms) with the records / without / what the library builds by itself")
for L in (20, 25, 30, 35, 40, 45, 50, 55, 62, 80):
    names = [s[:L] for s in joined]
    vec = HipTfidfVectorizer(dtype=np.float32, ctx=ctx)
    p = vec.prepare(names)
    vec.fit_prepared([p])
    A = vec.transform_prepared(p)
    r_, c_, nnz, _ = A.dims()
    out = []
    for q8 in ("1", "0", None):
        if q8 is not None:
            ctx.set_option("SG_Q8", q8)
        post = ctx.postings_build(A)
        best = None
        for _ in range(3):
            r = ctx.spgemm_topn(A, post, 10, thr, True)
            ctx.sync()
            st = ctx.stats()
            r.free()
            if best is None or st["ms_spgemm_topn"] < best["ms_spgemm_topn"]:
                best = st
        out.append(best)
        post.free()
        ctx.reset_options()
    assert out[0]["out_nnz"] == out[1]["out_nnz"] == out[2]["out_nnz"]
    print(f"cut at {L:3d} characters: {nnz / r_:5.1f} entries a row | " +
          " | ".join(f"{o['ms_spgemm_topn']:6.2f} ({o['ms_spgemm_kernel']:6.2f})" for o in out) +
          f" | candidates {out[1]['prune_survivors']:9d}, scored exactly with the filter {out[0]['prune_scored']:8d}", flush=True)
