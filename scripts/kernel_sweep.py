"""Timing sweep of the multiply (K3 + K4) over tile sizes / tile groups / occupancy, on the GPU box.
Development tool: prints one JSON line per configuration."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from string_grouper_amd import _native as N  # noqa: E402
from string_grouper_amd.synth import synth_names  # noqa: E402


def tfidf_device(ctx, names, dtype):
    from string_grouper_amd.vectorizer import HipTfidfVectorizer
    vec = HipTfidfVectorizer(dtype=dtype, ctx=ctx)
    t0 = time.perf_counter()
    p = vec.prepare(names)
    t1 = time.perf_counter()
    vec.fit_prepared([p])
    A = vec.transform_prepared(p)
    ctx.sync()
    t2 = time.perf_counter()
    print(json.dumps({"what": "vectorise", "n": len(names), "prepare_s": t1 - t0, "fit_transform_s": t2 - t1,
                      **{k: v for k, v in ctx.stats().items() if k.startswith("ms_")}}), flush=True)
    return A


def main():
    sizes = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "663000").split(",")]
    ctx = N.default_context(0)
    for n in sizes:
        names = synth_names(n, 1234)
        for dtype in (np.float32,):
            try:
                A = tfidf_device(ctx, names, dtype)
            except Exception as e:  # development sweep only: fall back to host-built input
                print(json.dumps({"what": "vectorise-failed", "err": repr(e)}), flush=True)
                from oracle import oracle as O
                (m,), _, _ = O.tfidf_sklearn(names, [names], dtype=dtype)
                A = ctx.csr_from_scipy(m)
            for tile in (1024, 2048):
                t0 = time.perf_counter()
                post = ctx.postings_build(A, tile)
                ctx.sync()
                t_post = time.perf_counter() - t0
                for group, depth in [(0, 8), (0, 16)]:
                    for wpc in (0,):
                        os.environ["SG_TILE_GROUP"] = str(group)
                        os.environ["SG_DEPTH"] = str(depth)
                        if wpc:
                            os.environ["SG_WAVES_PER_CU"] = str(wpc)
                        else:
                            os.environ.pop("SG_WAVES_PER_CU", None)
                        best = None
                        for rep in range(2):
                            t0 = time.perf_counter()
                            res = ctx.spgemm_topn(A, post, 10, 0.8, True)
                            ctx.sync()
                            dt = time.perf_counter() - t0
                            st = ctx.stats()
                            res.free()
                            best = dt if best is None else min(best, dt)
                        print(json.dumps({"what": "spgemm", "n": n, "dtype": np.dtype(dtype).name, "tile": tile,
                                          "group": group, "depth": depth, "wall_s": best, "ms_event": st["ms_spgemm_topn"],
                                          "ms_postings": st["ms_postings"], "post_wall_s": t_post,
                                          "macs": st["macs"], "bytes": st["spgemm_bytes"], "out_nnz": st["out_nnz"],
                                          "alg_TBps": st["spgemm_bytes"] / (st["ms_spgemm_topn"] * 1e-3) / 1e12,
                                          "rows_per_s": n / best}), flush=True)
                post.free()
            A.free()


if __name__ == "__main__":
    main()
