"""Self-join form vs one-sided form of the pruned multiply across sizes (development tool; fixes SG_SYM_MIN_ROWS)."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from string_grouper_amd import _native as N
from string_grouper_amd.synth import synth_names
from string_grouper_amd.vectorizer import HipTfidfVectorizer
ctx = N.default_context(0)
for n in (5000, 20000, 50000, 100000, 200000, 400000):
    names = synth_names(n, 1234)
    vec = HipTfidfVectorizer(dtype=np.float32, ctx=ctx)
    p = vec.prepare(names)
    vec.fit_prepared([p]); A = vec.transform_prepared(p); post = ctx.postings_build(A)
    out = {"n": n}
    for sym in ("1", "0"):
        ctx.set_option("SG_SYM", sym)        # (the library reads its switches when a context is created)
        best = 1e9
        for rep in range(3):
            r = ctx.spgemm_topn(A, post, 10, 0.8, True); ctx.sync()
            st = ctx.stats(); r.free()
            best = min(best, st["ms_spgemm_topn"])
        out["sym" if sym == "1" else "one_sided"] = round(best, 3)
        out["symmetric_ran" if sym == "1" else "_"] = st["prune_symmetric"]
    out.pop("_", None)
    print(json.dumps(out), flush=True)
    ctx.reset_options()
    post.free(); A.free()
