"""K1 / K2 timing probes (development tool): fit (tokenise + df atomics) for several replica counts, and the
tokenisation of a fresh handle of the same strings (no df atomics)."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from string_grouper_amd import _native as N
from string_grouper_amd.synth import synth_names
from string_grouper_amd.vectorizer import HipTfidfVectorizer
ctx = N.default_context(0)
n = int(os.environ.get("N", "663000"))
names = synth_names(n, 1234)
def ms(keys=("ms_tokenize", "ms_weight", "ms_vocab")):
    st = ctx.stats()
    return {k: round(st[k], 3) for k in keys}
for reps in ("8", "16", "32", "64"):
    os.environ["SG_DF_REPLICAS"] = reps
    for it in range(3):
        vec = HipTfidfVectorizer(dtype=np.float32, ctx=ctx)
        p = vec.prepare(names)
        vec.fit_prepared([p]); ctx.sync()
        fit_ms = ms()
        A = vec.transform_prepared(p); ctx.sync()
        cached_ms = ms()
        q = vec.prepare(names)          # a fresh handle: tokenised again, without the df atomics
        B = vec.transform_prepared(q); ctx.sync()
        fresh_ms = ms()
        A.free(); B.free()
    print(json.dumps({"n": n, "df_replicas": reps, "fit": fit_ms, "transform_cached": cached_ms, "transform_fresh": fresh_ms}), flush=True)
