"""Round 6: the forms outside the name-matching envelope at BASELINE's large size -- 5 M names self-join, top 10 at 0.6
(tile-by-tile form on its own index), top 100 at 0.8 (a row's own matches through the pair list), top 10 at 0.38 (exact
kernel, self-join form, own layout) -- every row against the one-sided exact kernel.   python scripts/big_forms_check.py [rows]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from string_grouper_amd import _native as N  # noqa: E402
from string_grouper_amd.synth import synth_names  # noqa: E402
from string_grouper_amd.vectorizer import HipTfidfVectorizer  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000000
ctx = N.default_context(0)
vec = HipTfidfVectorizer(dtype=np.float32, ctx=ctx)
p = vec.prepare(synth_names(n, 4321))
vec.fit_prepared([p])
A = vec.transform_prepared(p)
post = ctx.postings_build(A)
for top_n, thr in ((10, 0.6), (100, 0.8), (10, 0.38)):
    t0 = time.perf_counter()
    res = ctx.spgemm_topn(A, post, top_n, thr, True)
    ctx.sync()
    t1 = time.perf_counter()
    st = ctx.stats()
    ctx.set_option("SG_PRUNE", "0")
    ctx.set_option("SG_EXACT_SYM", "0")
    ref = ctx.spgemm_topn(A, post, top_n, thr, True)
    ctx.sync()
    t2 = time.perf_counter()
    ctx.reset_options()
    ok = True
    cg, cr = res.counts(), ref.counts()
    ok = np.array_equal(cg, cr)
    if ok:                       # row blocks: the full results are 2 x 5 M x top_n x 8 bytes
        g, w = res.to_host(), ref.to_host()
        mask = np.arange(w[0].shape[1])[None, :] < w[2][:, None]
        ok = bool(np.array_equal(g[0][mask], w[0][mask]) and np.array_equal(g[1][mask], w[1][mask]))
        del g, w, mask
    print(f"{n} names, top {top_n} at {thr}: {1e3 * (t1 - t0):8.1f} ms (first call: indexes of the form included), multiply "
          f"{st['ms_spgemm_topn']:8.1f} ms, self-join form {st['prune_symmetric']}, pruned rows {st['prune_rows']}, matches {st['out_nnz']}; "
          f"one-sided exact kernel {1e3 * (t2 - t1):8.1f} ms; identical: {ok}", flush=True)
    assert ok
    res.free()
    ref.free()
