"""BASELINE.json configs[3] and configs[4] at FULL size on ONE MI355X (the tests run one of eight row blocks):
5 M names self-join (top 10, 0.8) and 10 M x 1 M master x duplicates (top 20, 0.7), fp32.  Kernel times from sg_stats,
size-independent properties of the whole result and sampled rows against the CPU port (tests/test_parity_gpu.py).

    python scripts/full_configs.py [3] [4] [exact]

``exact``: additionally ALL rows of the result are compared, bit for bit, with the exact kernel K4 run on every row of the
same matrices (no pruning, no grouping of identical rows: SG_PRUNE=0, SG_COLLAPSE=0) -- the kernel the sampled rows and
the smaller sizes pin on the CPU port.  (The port itself on all rows of 5 M x 5 M is ~ 7e12 products: a quarter of an
hour on the box's 16 host cores.)
"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from string_grouper_amd import _native as N  # noqa: E402
from string_grouper_amd.synth import synth_names  # noqa: E402
from string_grouper_amd.vectorizer import HipTfidfVectorizer  # noqa: E402
from tests.test_parity_gpu import _check_slice_properties  # noqa: E402


def line(tag, st, t_wall, extra=""):
    print(f"{tag}: tokenise {st['ms_tokenize']:.1f} + vocab {st['ms_vocab']:.1f} + weight {st['ms_weight']:.1f} + postings "
          f"{st['ms_postings']:.1f} + multiply {st['ms_spgemm_topn']:.1f} ms (kernel {st['ms_spgemm_kernel']:.1f}); self-join form "
          f"{st['prune_symmetric']}, pruned rows {st['prune_rows']}, rows for the exact kernel {st['exact_rows']}, postings streamed "
          f"{st['prune_postings']:.3e} of {st['macs']:.3e} products, candidates {st['prune_survivors']:.3e}, pairs scored exactly {st['prune_scored']:.3e}, matches {st['out_nnz']}; "
          f"wall {t_wall:.2f} s {extra}", flush=True)


def same_as_exact_kernel(ctx, res, A, B, top_n, thr, what):
    """Every row of `res` against the exact kernel on all rows."""
    ctx.set_option("SG_PRUNE", "0")
    ctx.set_option("SG_COLLAPSE", "0")
    t0 = time.time()
    post = ctx.postings_build(B)
    ex = ctx.spgemm_topn(A, post, top_n, thr, True)
    ctx.sync()
    st = ctx.stats()
    ctx.reset_options()
    assert st["prune_rows"] == 0
    c0, v0, n0 = res.to_host()
    c1, v1, n1 = ex.to_host()
    ex.free()
    post.free()
    mask = np.arange(c0.shape[1])[None, :] < n0[:, None]
    same = bool(np.array_equal(n0, n1) and np.array_equal(c0[mask], c1[mask]) and np.array_equal(v0[mask], v1[mask]))
    print(f"{what}: ALL {len(n0)} rows ({int(n0.sum())} matches) against the exact kernel on every row "
          f"({st['macs']:.3e} products, {st['ms_spgemm_topn']:.0f} ms; {time.time() - t0:.0f} s with the index and the copies): "
          f"{'identical, bit for bit' if same else 'DIFFERENT'}", flush=True)
    assert same, what


def config3(ctx, exact=False):
    n = 5_000_000
    t0 = time.time()
    names = synth_names(n, 1234)
    print(f"config 3: {n} names generated in {time.time() - t0:.0f} s", flush=True)
    vec = HipTfidfVectorizer(dtype=np.float32, ctx=ctx)
    p = vec.prepare(names)
    for rep in range(2):
        t0 = time.time()
        vec.fit_prepared([p])
        A = vec.transform_prepared(p)
        post = ctx.postings_build(A)
        res = ctx.spgemm_topn(A, post, 10, 0.8, True)
        st = ctx.stats()
        line(f"config 3 (5 M self-join, top 10, 0.8) run {rep}", st, time.time() - t0)
        if rep == 0:
            for h in (res, post, A):
                h.free()
    A_host = A.to_scipy()
    _check_slice_properties(res, 0, n, n, 10, 0.8, True, A_host, A_host, 400, "config3 full")
    print("config 3: properties of all rows + 400 sampled rows equal the CPU port", flush=True)
    del A_host
    if exact:
        same_as_exact_kernel(ctx, res, A, A, 10, 0.8, "config 3")
    for h in (res, post, A):
        h.free()
    ctx.trim()


def config4(ctx, exact=False):
    n_m, n_d = 10_000_000, 1_000_000
    t0 = time.time()
    master = synth_names(n_m, 1234)
    dupes = synth_names(n_d, seed=4321, perturb_of=master, perturb_frac=0.5)
    print(f"config 4: names generated in {time.time() - t0:.0f} s", flush=True)
    vec = HipTfidfVectorizer(dtype=np.float32, ctx=ctx)
    pm, pd_ = vec.prepare(master), vec.prepare(dupes)
    del master
    for rep in range(2):
        t0 = time.time()
        vec.fit_prepared([pm, pd_])
        A = vec.transform_prepared(pm)
        B = vec.transform_prepared(pd_)
        post = ctx.postings_build(B)
        res = ctx.spgemm_topn(A, post, 20, 0.7, True)
        st = ctx.stats()
        line(f"config 4 (10 M x 1 M, top 20, 0.7) run {rep}", st, time.time() - t0)
        if rep == 0:
            for h in (res, post, A, B):
                h.free()
    A_host, B_host = A.to_scipy(), B.to_scipy()
    _check_slice_properties(res, 0, n_m, n_d, 20, 0.7, False, A_host, B_host, 400, "config4 full")
    print("config 4: properties of all rows + 400 sampled rows equal the CPU port", flush=True)
    del A_host, B_host
    if exact:
        same_as_exact_kernel(ctx, res, A, B, 20, 0.7, "config 4")
    for h in (res, post, A, B):
        h.free()
    ctx.trim()


if __name__ == "__main__":
    args = sys.argv[1:]
    exact = "exact" in args
    which = [a for a in args if a != "exact"] or ["3", "4"]
    ctx = N.Context()
    if "3" in which:
        config3(ctx, exact)
    if "4" in which:
        config4(ctx, exact)
