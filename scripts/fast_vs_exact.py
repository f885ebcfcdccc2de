"""Fast path vs exact kernel on the bench workload: identical results + timing.  Development tool."""
import os, sys, json, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from string_grouper_amd import _native as N
from string_grouper_amd.synth import synth_names
from string_grouper_amd.vectorizer import HipTfidfVectorizer

n = int(sys.argv[1]) if len(sys.argv) > 1 else 663000
ctx = N.default_context(0)
names = synth_names(n, 1234)
for dtype in (np.float32, np.float64):
    vec = HipTfidfVectorizer(dtype=dtype, ctx=ctx); p = vec.prepare(names); vec.fit_prepared([p]); A = vec.transform_prepared(p)
    post = ctx.postings_build(A)
    out = {}
    for mode in ("fast", "exact"):
        os.environ["SG_EXACT_ONLY"] = "1" if mode == "exact" else "0"
        for rep in range(2):
            r = ctx.spgemm_topn(A, post, 10, 0.8, True); ctx.sync(); st = ctx.stats()
            if rep == 0: r.free()
        out[mode] = (r.to_host(), st["ms_spgemm_topn"])
        r.free()
    (c1, v1, n1), t1 = out["fast"]; (c2, v2, n2), t2 = out["exact"]
    mask = np.arange(c1.shape[1])[None, :] < n1[:, None]
    same = np.array_equal(n1, n2) and np.array_equal(c1[mask], c2[mask]) and np.array_equal(v1[mask], v2[mask])
    print(json.dumps({"n": n, "dtype": np.dtype(dtype).name, "fast_ms": t1, "exact_ms": t2, "identical": bool(same),
                      "matches": int(n1.sum())}), flush=True)
    post.free(); A.free()
