"""BASELINE.json configs[3] and configs[4] at FULL size on ONE MI355X (the tests run one of eight row blocks):
5 M names self-join (top 10, 0.8) and 10 M x 1 M master x duplicates (top 20, 0.7), fp32.  Kernel times from sg_stats,
size-independent properties of the whole result and sampled rows against the CPU port (tests/test_parity_gpu.py).

    python scripts/full_configs.py [3] [4]
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
          f"{st['prune_postings']:.3e} of {st['macs']:.3e} products, pairs scored {st['prune_survivors']:.3e}, matches {st['out_nnz']}; "
          f"wall {t_wall:.2f} s {extra}", flush=True)


def config3(ctx):
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
    for h in (res, post, A):
        h.free()
    ctx.trim()


def config4(ctx):
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
    for h in (res, post, A, B):
        h.free()
    ctx.trim()


if __name__ == "__main__":
    which = sys.argv[1:] or ["3", "4"]
    ctx = N.Context()
    if "3" in which:
        config3(ctx)
    if "4" in which:
        config4(ctx)
