"""Performance of the hot path on inputs that are NOT shuffled synthetic names: the same 663 k names sorted
alphabetically (real lists often are), sorted by length, with a tenth of the list replaced by one repeated name (a hub),
and all names distinct (no near-duplicates).  Kernel times from sg_stats; result compared with the shuffled run where
the multiset of names is the same (match count)."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from string_grouper_amd import _native as N  # noqa: E402
from string_grouper_amd.synth import synth_names  # noqa: E402
from string_grouper_amd.vectorizer import HipTfidfVectorizer  # noqa: E402


def run(ctx, names, tag, top_n=10, thr=0.8):
    vec = HipTfidfVectorizer(dtype=np.float32, ctx=ctx)
    p = vec.prepare(names)
    best = None
    for rep in range(3):
        t0 = time.perf_counter()
        vec.fit_prepared([p])
        A = vec.transform_prepared(p)
        post = ctx.postings_build(A)
        res = ctx.spgemm_topn(A, post, top_n, thr, True)
        ctx.sync()
        wall = (time.perf_counter() - t0) * 1e3
        st = ctx.stats()
        if best is None or wall < best[0]:
            best = (wall, st)
        for h in (res, post, A):
            h.free()
    wall, st = best
    print(f"{tag:34s} wall {wall:7.2f} ms: tokenise {st['ms_tokenize']:.2f} vocab {st['ms_vocab']:.2f} weight {st['ms_weight']:.2f} "
          f"postings {st['ms_postings']:.2f} multiply {st['ms_spgemm_topn']:.2f} (kernel {st['ms_spgemm_kernel']:.2f}); self-join form "
          f"{st['prune_symmetric']}, exact rows {st['exact_rows']}, postings {st['prune_postings']:.3e}, pairs scored "
          f"{st['prune_survivors']:.3e}, matches {st['out_nnz']}", flush=True)


def run_pair(ctx, master, dupes, tag, top_n=20, thr=0.7):
    vec = HipTfidfVectorizer(dtype=np.float32, ctx=ctx)
    pm, pd_ = vec.prepare(master), vec.prepare(dupes)
    best = None
    for rep in range(3):
        t0 = time.perf_counter()
        vec.fit_prepared([pm, pd_])
        A = vec.transform_prepared(pm)
        B = vec.transform_prepared(pd_)
        post = ctx.postings_build(B)
        res = ctx.spgemm_topn(A, post, top_n, thr, True)
        ctx.sync()
        wall = (time.perf_counter() - t0) * 1e3
        st = ctx.stats()
        if best is None or wall < best[0]:
            best = (wall, st)
        for h in (res, post, A, B):
            h.free()
    wall, st = best
    print(f"{tag:34s} wall {wall:7.2f} ms: postings {st['ms_postings']:.2f} multiply {st['ms_spgemm_topn']:.2f}; exact rows {st['exact_rows']}, "
          f"postings {st['prune_postings']:.3e}, pairs scored {st['prune_survivors']:.3e}, matches {st['out_nnz']}", flush=True)


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 663000
    ctx = N.Context()
    names = synth_names(n, 1234)
    run(ctx, names, "shuffled (the bench workload)")
    if len(sys.argv) <= 2:
        run(ctx, sorted(names), "sorted alphabetically")
        run(ctx, sorted(names, key=len), "sorted by length")
        run(ctx, sorted(names, reverse=True), "sorted descending")
        hub = list(names)
        for i in range(0, n, 10):
            hub[i] = "ACME HOLDINGS INTERNATIONAL LLC"
        run(ctx, hub, "a tenth of the rows one name")
        run(ctx, synth_names(n, 99, perturb_frac=0.0), "no near-duplicates")
        run(ctx, [s.lower() + " " + s[:3] for s in names], "lower case + a repeated prefix")
    if len(sys.argv) > 2:
        m = synth_names(1_000_000, 5)
        d = synth_names(300_000, 6, perturb_of=m, perturb_frac=0.5)
        run_pair(ctx, m, d, "1 M x 300 k, shuffled")
        run_pair(ctx, sorted(m), sorted(d), "1 M x 300 k, both sorted")
        for k in (2000, 10000, 50000, 100000):
            run(ctx, names[:k], f"{k} names")
