import sys, time
import numpy as np
sys.path.insert(0, ".")
from string_grouper_amd import _native as N
from string_grouper_amd.synth import synth_names
from string_grouper_amd.vectorizer import HipTfidfVectorizer
n = 5000000
ctx = N.default_context(0)
vec = HipTfidfVectorizer(dtype=np.float32, ctx=ctx)
p = vec.prepare(synth_names(n, 4321))
vec.fit_prepared([p])
A = vec.transform_prepared(p)
post = ctx.postings_build(A)
for top_n, thr in ((10, 0.5), (10, 0.45), (10, 0.42)):
    for rep in range(2):
        t0 = time.perf_counter()
        res = ctx.spgemm_topn(A, post, top_n, thr, True)
        ctx.sync()
        t1 = time.perf_counter()
        st = ctx.stats()
        res.free()
    print(f"5 M names, top {top_n} at {thr}: {1e3*(t1-t0):9.1f} ms, self-join form {st['prune_symmetric']}, pruned rows {st['prune_rows']}, matches {st['out_nnz']}", flush=True)
