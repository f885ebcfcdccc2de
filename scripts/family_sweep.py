"""K4p against K4 on data families other than the 663k headline: n-gram sizes 2 / 3 / 4, short and long strings, small
vocabularies, master x duplicates.  For each: identical results (asserted), kernel times from sg_stats, what the pruning
streamed.  Run on the GPU box:  python scripts/family_sweep.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from string_grouper_amd import _native as N  # noqa: E402
from string_grouper_amd.synth import synth_names  # noqa: E402
from string_grouper_amd.vectorizer import HipTfidfVectorizer  # noqa: E402

ctx = N.default_context(0)
OPTIONS = dict(a.split("=", 1) for a in sys.argv[1:] if "=" in a and not a.startswith("only="))   # e.g. SG_PRUNE_MIN_THRESHOLD=0.3
ONLY = [a[5:] for a in sys.argv[1:] if a.startswith("only=")]
rng = np.random.default_rng(1)


def long_names(n, seed):
    base = synth_names(n * 3, seed)
    return [" ".join(base[3 * i:3 * i + 3]) for i in range(n)]        # ~75 characters: rows above 64 non-zeros


def run(label, master, dups, top_n, thr, **kw):
    if ONLY and not any(o in label for o in ONLY):
        return
    vec = HipTfidfVectorizer(dtype=np.float32, ctx=ctx, **kw)
    pm = vec.prepare(master)
    sets = [pm] + ([vec.prepare(dups)] if dups is not None else [])
    vec.fit_prepared(sets)
    A = vec.transform_prepared(pm)
    B = A if dups is None else vec.transform_prepared(sets[1])
    post = ctx.postings_build(B)
    out = {}
    for mode, env in (("pruned", {}), ("one-sided", {"SG_SYM": "0"}), ("exact", {"SG_PRUNE": "0"})):
        for k, v in list(env.items()) + list(OPTIONS.items()):
            ctx.set_option(k, v)          # (the library reads its switches when a context is created)
        if mode == "exact":
            post.free()
            post = ctx.postings_build(B)
        best = None
        for _ in range(2):
            r = ctx.spgemm_topn(A, post, top_n, thr, True)
            ctx.sync()
            st = ctx.stats()
            if best is None or st["ms_spgemm_topn"] < best["ms_spgemm_topn"]:
                best = st
            res = r.to_host()
            r.free()
        out[mode] = (best, res)
        ctx.reset_options()
    ref = out["exact"][1]
    for mode in ("pruned", "one-sided"):
        got = out[mode][1]
        mask = np.arange(ref[0].shape[1])[None, :] < ref[2][:, None]
        assert np.array_equal(got[2], ref[2]) and np.array_equal(got[0][mask], ref[0][mask]) and \
            np.array_equal(got[1][mask], ref[1][mask]), (label, mode)
    r_, c_, nnz, _ = A.dims()
    p, o, e = out["pruned"][0], out["one-sided"][0], out["exact"][0]
    print(f"{label:46s} rows {r_:7d} V {c_:6d} nnz/row {nnz / max(r_, 1):5.1f}  K4p {p['ms_spgemm_topn']:8.2f} ms (sym={p['prune_symmetric']}, "
          f"to-exact {p['exact_rows']})  one-sided {o['ms_spgemm_topn']:8.2f}  K4 {e['ms_spgemm_topn']:8.2f}  streamed "
          f"{100.0 * o['prune_postings'] / max(o['macs'], 1):5.1f}% of {o['macs']:.3g} MACs, survivors {o['prune_survivors']}", flush=True)
    post.free()


t0 = time.time()
n = 200000
names = synth_names(n, 77)
run("SynthNames 200k 3-grams ntop10 0.8", names, None, 10, 0.8)
run("SynthNames 200k 2-grams ntop10 0.8", names, None, 10, 0.8, ngram_size=2)
run("SynthNames 200k 4-grams ntop10 0.8", names, None, 10, 0.8, ngram_size=4)
run("SynthNames 200k 3-grams ntop20 0.6", names, None, 20, 0.6)
run("SynthNames 200k 3-grams ntop5 0.9", names, None, 5, 0.9)
run("SynthNames 200k 3-grams ntop100 0.8", names, None, 100, 0.8)    # 65 .. 128: pruned, rows with full lists to the exact kernel
run("SynthNames 200k 3-grams ntop100 0.6", names, None, 100, 0.6)
for low in (0.45, 0.4, 0.35, 0.3):       # below 0.45 the pruned kernel only runs with SG_PRUNE_MIN_THRESHOLD=<lower>
    run(f"low threshold: 200k 3-grams ntop10 {low}", names, None, 10, low)
run("long names (3 joined) 100k 3-grams ntop10 0.8", long_names(100000, 5), None, 10, 0.8)
def very_long(n, k, seed):
    base = synth_names(n * k, seed)
    return [" ".join(base[k * i:k * i + k]) for i in range(n)]


run("long names (5 joined) 50k 3-grams ntop10 0.8", very_long(50000, 5, 6), None, 10, 0.8)       # ~100 entries: the wide launch
run("very long (8 joined) 50k 3-grams ntop10 0.8", very_long(50000, 8, 7), None, 10, 0.8)        # ~160 entries: beyond the pruned kernel
digits = ["%09d" % int(x) for x in rng.integers(0, 10 ** 9, 200000)]
run("9-digit numbers 200k 3-grams (V<=1000) 0.8", digits, None, 10, 0.8)
m = synth_names(300000, 3)
d = synth_names(100000, 4, perturb_of=m, perturb_frac=0.5)
run("master 300k x duplicates 100k ntop20 0.7", m, d, 20, 0.7)
print("total %.1f s" % (time.time() - t0))
