"""Is K4 bound by the memory side?  Same left matrix (663k rows), right-hand sides of growing size:
if the MAC rate drops as the postings outgrow the 4 MiB per-XCD L2 / the 256 MiB Infinity Cache, it is."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from string_grouper_amd import _native as N
from string_grouper_amd.synth import synth_names
from string_grouper_amd.vectorizer import HipTfidfVectorizer
ctx = N.default_context(0)
names = synth_names(663000, 1234)
vec = HipTfidfVectorizer(dtype=np.float32, ctx=ctx); p = vec.prepare(names); vec.fit_prepared([p]); A = vec.transform_prepared(p)
for mode in ("exact", "fast"):
    os.environ["SG_EXACT_ONLY"] = "1" if mode == "exact" else "0"
    for nr in (8192, 20000, 50000, 100000, 200000, 400000, 663000):
        B = A.row_block(0, nr)
        post = ctx.postings_build(B)
        for rep in range(2):
            r = ctx.spgemm_topn(A, post, 10, 0.8, True); ctx.sync(); st = ctx.stats(); r.free()
        print(json.dumps({"mode": mode, "n_right": nr, "postings_MB": round(nr * 19 * 8 / 1e6, 1), "ms": round(st["ms_spgemm_topn"], 2),
                          "GMAC_per_s": round(st["macs"] / st["ms_spgemm_topn"] / 1e6, 1)}), flush=True)
        post.free(); B.free()
