import sys, numpy as np
sys.path.insert(0, ".")
from string_grouper_amd import _native as N
from string_grouper_amd.synth import synth_names
from string_grouper_amd.vectorizer import HipTfidfVectorizer
ctx = N.Context()
names = synth_names(663000, 1234)
vec = HipTfidfVectorizer(dtype=np.float32, ctx=ctx)
p = vec.prepare(names)
for rep in range(4):
    vec.fit_prepared([p]); A = vec.transform_prepared(p); post = ctx.postings_build(A)
    res = ctx.spgemm_topn(A, post, 10, 0.8, True); ctx.sync()
    for h in (res, post, A): h.free()
