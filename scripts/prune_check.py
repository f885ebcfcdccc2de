"""Pruned vs exact multiply on the GPU box: bit-equality of the results and timing.  Development tool."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from string_grouper_amd import _native as N  # noqa: E402
from string_grouper_amd.synth import synth_names  # noqa: E402
from string_grouper_amd.vectorizer import HipTfidfVectorizer  # noqa: E402


def run(ctx, A, top_n, thr, prune, delta=None, reps=2, freq=None, tile=None):
    if tile is not None:
        os.environ["SG_PRUNE_TILE"] = str(tile)
    os.environ["SG_PRUNE"] = "1" if prune else "0"
    if delta is not None:
        os.environ["SG_PRUNE_DELTA"] = str(delta)
    if freq is not None:
        os.environ["SG_PRUNE_FREQ"] = str(freq)
    post = ctx.postings_build(A, 0)
    best, out, st = None, None, None
    for _ in range(reps):
        ctx.sync()
        t0 = time.perf_counter()
        res = ctx.spgemm_topn(A, post, top_n, thr, True)
        ctx.sync()
        dt = time.perf_counter() - t0
        st = ctx.stats()
        if out is None:
            out = res.to_host()
        res.free()
        best = dt if best is None else min(best, dt)
    post.free()
    return best, out, st


def main():
    sizes = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "20000,200000,663000").split(",")]
    deltas = [float(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "0.2").split(",")]
    dtypes = [np.float32, np.float64] if (len(sys.argv) > 3 and sys.argv[3] == "both") else [np.float32]
    freqs = [float(x) for x in (sys.argv[4] if len(sys.argv) > 4 else "0.003").split(",")]
    tiles = [int(x) for x in (sys.argv[5] if len(sys.argv) > 5 else "12").split(",")]
    thresholds = [float(x) for x in (sys.argv[6] if len(sys.argv) > 6 else "0.8").split(",")]
    ctx = N.default_context(0)
    for n in sizes:
        names = synth_names(n, 1234)
        for dtype in dtypes:
            vec = HipTfidfVectorizer(dtype=dtype, ctx=ctx)
            p = vec.prepare(names)
            vec.fit_prepared([p])
            A = vec.transform_prepared(p)
            for top_n, thr in [(10, t) for t in thresholds]:
                t_ex, o_ex, st_ex = run(ctx, A, top_n, thr, False)
                print(json.dumps({"n": n, "dtype": np.dtype(dtype).name, "mode": "exact", "thr": thr, "ms": st_ex["ms_spgemm_topn"],
                                  "wall_ms": t_ex * 1e3, "macs": st_ex["macs"], "out": st_ex["out_nnz"]}), flush=True)
                for tile, delta, freq in [(t, d, f) for t in tiles for f in freqs for d in deltas]:
                    t_pr, o_pr, st_pr = run(ctx, A, top_n, thr, True, delta, freq=freq, tile=tile)
                    mask = np.arange(o_ex[0].shape[1])[None, :] < o_ex[2][:, None]
                    same = (np.array_equal(o_ex[2], o_pr[2]) and np.array_equal(o_ex[0][mask], o_pr[0][mask])
                            and np.array_equal(o_ex[1][mask], o_pr[1][mask]))
                    if not same:
                        cnt_e, cnt_p = o_ex[2], o_pr[2]
                        bad = np.nonzero(cnt_e != cnt_p)[0]
                        print(json.dumps({"count_mismatch_rows": int(len(bad)), "first": bad[:5].tolist(),
                                          "cnt_exact": cnt_e[bad[:5]].tolist(), "cnt_pruned": cnt_p[bad[:5]].tolist()}))
                    print(json.dumps({"n": n, "dtype": np.dtype(dtype).name, "mode": "pruned", "thr": thr, "tile": tile, "delta": delta, "freq": freq,
                                      "ms": st_pr["ms_spgemm_topn"], "wall_ms": t_pr * 1e3, "identical": bool(same),
                                      "rows": st_pr["prune_rows"], "postings": st_pr["prune_postings"],
                                      "survivors": st_pr["prune_survivors"], "exact_rows": st_pr["exact_rows"],
                                      "out": st_pr["out_nnz"]}), flush=True)
            A.free()


if __name__ == "__main__":
    main()
