"""One share of the self-join form, several times, for a kernel trace: which launch of the share takes how long.
python scripts/share_trace.py [world=2] [rank=0] [rows=663000]   (run under rocprofv3 --kernel-trace --stats)"""
import sys

import numpy as np

sys.path.insert(0, ".")
from string_grouper_amd import _native as N  # noqa: E402
from string_grouper_amd import distributed as D  # noqa: E402
from string_grouper_amd.synth import synth_names  # noqa: E402
from string_grouper_amd.vectorizer import HipTfidfVectorizer  # noqa: E402

world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
rank = int(sys.argv[2]) if len(sys.argv) > 2 else 0
n = int(sys.argv[3]) if len(sys.argv) > 3 else 663000
ctx = N.Context()
vec = HipTfidfVectorizer(dtype=np.float32, ctx=ctx)
p = vec.prepare(synth_names(n, 1234))
vec.fit_prepared([p])
A = vec.transform_prepared(p)
post = ctx.postings_build(A)
n_index = ctx.postings_rows(post)[0]
lo, hi, step = (0, n_index, 1) if world == 1 else D.selfjoin_share(n_index, rank, world)
for rep in range(5):
    got = ctx.selfjoin_range(A, post, 10, 0.8, lo, hi, step)
    ctx.sync()
    print(rep, ctx.stats()["ms_spgemm_kernel"], flush=True)
    got[0].free()
    ctx.device_free(got[1])
