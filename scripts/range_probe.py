"""How long does the slowest single row of the self-join pass take?  Ranges of 4096 rows (one per wave: every wave takes a
helping of four, so only a quarter of the waves work) and of 16384 rows at several positions of the 663 k job: the kernel
time of such a launch is about the duration of its slowest helping -- the tail every multi-GPU range ends with
(scripts/sim_scaling.py).  python scripts/range_probe.py [rows=663000]"""
import sys

import numpy as np

sys.path.insert(0, ".")
from string_grouper_amd import _native as N  # noqa: E402
from string_grouper_amd.synth import synth_names  # noqa: E402
from string_grouper_amd.vectorizer import HipTfidfVectorizer  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 663000
ctx = N.Context()
names = synth_names(n, 1234)
vec = HipTfidfVectorizer(dtype=np.float32, ctx=ctx)
p = vec.prepare(names)
vec.fit_prepared([p])
A = vec.transform_prepared(p)
post = ctx.postings_build(A)
n = ctx.postings_rows(post)[0]      # (rows of the index: groups of identical rows when the library grouped them)
print(f"index over {n} rows")
for width in (256, 4096, 16384, 65536):
    out = []
    for frac in (0.1, 0.3, 0.5, 0.7, 0.9, 1.0):
        hi = int(frac * n)
        lo = max(0, hi - width)
        best = None
        for rep in range(3):
            got = ctx.selfjoin_range(A, post, 10, 0.8, lo, hi)
            ctx.sync()
            k = ctx.stats()["ms_spgemm_kernel"]
            got[0].free()
            ctx.device_free(got[1])
            best = k if best is None or k < best else best
        out.append((frac, round(best, 3)))
    print(f"ranges of {width} rows ending at fraction f of the list: kernel ms {out}")
