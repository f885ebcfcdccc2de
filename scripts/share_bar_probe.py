"""How a share's pass 1 of the self-join form depends on the bar from which a row is set aside for the launch over parts
(SG_HEAVY_ROUNDS): python scripts/share_bar_probe.py [world=8] [rows=5000000] [ranks=0,6]"""
import sys

import numpy as np

sys.path.insert(0, ".")
from string_grouper_amd import _native as N  # noqa: E402
from string_grouper_amd import distributed as D  # noqa: E402
from string_grouper_amd.synth import synth_names  # noqa: E402
from string_grouper_amd.vectorizer import HipTfidfVectorizer  # noqa: E402

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5000000
ranks = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "0,6").split(",")]
ctx = N.Context()
vec = HipTfidfVectorizer(dtype=np.float32, ctx=ctx)
p = vec.prepare(synth_names(n, 1234))
vec.fit_prepared([p])
A = vec.transform_prepared(p)
post = ctx.postings_build(A)
n_index = ctx.postings_rows(post)[0]
for bar in (None, "0", "200", "400", "800", "1600", "3200"):
    ctx.set_option("SG_HEAVY_ROUNDS", bar)
    line = []
    for rank in ranks:
        lo, hi, step = D.selfjoin_share(n_index, rank, world)
        best = 1e9
        for rep in range(3):
            got = ctx.selfjoin_range(A, post, 10, 0.8, lo, hi, step)
            ctx.sync()
            best = min(best, ctx.stats()["ms_spgemm_kernel"])
            got[0].free()
            ctx.device_free(got[1])
        line.append(f"rank {rank}: {best:7.2f} ms")
    print(f"SG_HEAVY_ROUNDS={bar}: " + "   ".join(line), flush=True)
