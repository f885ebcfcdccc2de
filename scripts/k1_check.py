"""K1/K2/K3 timing + parity of the vectoriser against sklearn (development tool)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from string_grouper_amd import _native as N
from string_grouper_amd.synth import synth_names
from string_grouper_amd.vectorizer import HipTfidfVectorizer
from oracle import oracle as O
ctx = N.default_context(0)
import itertools
for n, rep_env in itertools.product((50000, 663000), ("8",)):
    os.environ["SG_DF_REPLICAS"] = rep_env
    names = synth_names(n, 1234)
    vec = HipTfidfVectorizer(dtype=np.float32, ctx=ctx)
    p = vec.prepare(names)
    for rep in range(3):
        vec = HipTfidfVectorizer(dtype=np.float32, ctx=ctx)
        vec.fit_prepared([p]); A = vec.transform_prepared(p); post = ctx.postings_build(A); ctx.sync()
        st = ctx.stats()
        if rep < 2:
            post.free(); A.free()
    out = {"n": n, "df_replicas": rep_env, **{k: round(v, 3) for k, v in st.items() if k in ("ms_tokenize", "ms_weight", "ms_postings")}}
    if n <= 50000:
        (m,), _, _ = O.tfidf_sklearn(names, [names], dtype=np.float32)
        h = A.to_scipy()
        out["tfidf_identical"] = bool(np.array_equal(h.indptr, m.indptr) and np.array_equal(h.indices, m.indices) and np.array_equal(h.data, m.data))
    print(json.dumps(out), flush=True)
