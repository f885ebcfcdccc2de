"""Projected strong-scaling curve of the headline job with MEASURED compute.

No multi-GPU node is available to the builder; the driver measures the real curve at round end.  What one GPU can show is
every rank's share of the work, run one rank at a time on the same device with the same library calls the N-rank driver
makes (string_grouper_amd/distributed.py: distributed_self_join in its self-join form over row ranges):

  per rank r of N   tokenise + weight its row block (K1, K2 -- measured on block r of the list)
                    inverted index of the WHOLE matrix (K3 -- replicated work, measured once)
                    pass 1 of the self-join form over ITS share of the positions (sg_selfjoin_range -- measured per rank;
                    distributed.selfjoin_share: interleaved by default, SG_DIST_INTERLEAVE=0 for contiguous ranges)
                    merge of the mirrored pairs that point into its range (sg_selfjoin_merge) and, when the index is one
                    over groups of identical rows, expansion of its groups into their member rows
                    (sg_topn_expand_groups) -- measured per rank through HipOps.selfjoin_merge, the driver's call
  collectives       all-reduce of the dense df table, all-gather of the CSR blocks, all-gather of the mirrored pairs:
                    bytes a rank receives / 300 GB/s (7 xGMI links x ~153 GB/s point to point, ring collectives are
                    per-link bound: MI355X_MICROARCH.md) + 20 us per collective -- a MODEL, labelled as such
  host round trips  measured: wall-clock of a rank's sequence minus the device time of its kernels

Critical path of N ranks = max over ranks of (fixed + range compute + merge) + collectives.  Prints one line per N and a
JSON summary (committed under profiles/).

    python scripts/sim_scaling.py [rows=663000] [dtype=f32]
"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from string_grouper_amd import _native as N  # noqa: E402
from string_grouper_amd import distributed as D  # noqa: E402
from string_grouper_amd.synth import synth_names  # noqa: E402
from string_grouper_amd.vectorizer import HipTfidfVectorizer  # noqa: E402

LINK_GBPS = 300.0
COLLECTIVE_LATENCY_MS = 0.02


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 663000
    dtype = np.float64 if (len(sys.argv) > 2 and sys.argv[2] == "f64") else np.float32
    s = 8 if dtype == np.float64 else 4
    ctx = N.Context()
    names = synth_names(n, 1234)
    make_vec = lambda: HipTfidfVectorizer(dtype=dtype, ctx=ctx)  # noqa: E731

    def ms(fn, reps=3):
        """(best wall-clock ms, kernel ms by group of the last call)"""
        best = None
        for _ in range(reps):
            ctx.sync()
            t0 = time.perf_counter()
            out = fn()
            ctx.sync()
            t = (time.perf_counter() - t0) * 1e3
            best = t if best is None or t < best else best
            st = ctx.stats()
        return best, {k[3:]: v for k, v in st.items() if k.startswith("ms_")}, out

    # the whole matrix once (what every rank holds after the CSR all-gather) and its index (replicated work)
    vec = make_vec()
    prepared = vec.prepare(names)
    vec.fit_prepared([prepared])
    A = vec.transform_prepared(prepared)
    nnz = A.dims()[2]
    t_post, k_post, post = ms(lambda: ctx.postings_build(A), reps=1)
    t_post, k_post, _p2 = ms(lambda: ctx.postings_build(A))
    _p2.free()
    report = {"rows": n, "dtype": "f64" if s == 8 else "f32", "nnz": int(nnz), "postings_ms_wall": t_post,
              "postings_ms_kernels": k_post.get("postings", 0.0), "ranks": {}}
    print(f"# {n} rows, nnz {nnz}; inverted index of the whole matrix: {t_post:.3f} ms wall ({k_post.get('postings', 0):.3f} ms kernels)")
    one_gpu = None
    ops = D.HipOps(ctx, make_vec)
    n_index, _, grouped = ctx.postings_rows(post)      # (an index over groups of identical rows: ranges of groups)
    report["index_rows"] = int(n_index)
    print(f"# index over {n_index} rows" + (" (groups of identical rows)" if grouped else ""))
    # warm-up of the driver's calls (torch's first kernels, the library's pools): one whole pass + merge, thrown away
    w = ops.selfjoin_range(A, post, 10, 0.8, 0, n_index)
    ops.selfjoin_merge(w, ops.selfjoin_pairs(w).clone(), 0, n_index).free()
    worlds = tuple(int(x) for x in sys.argv[3].split(",")) if len(sys.argv) > 3 else (1, 2, 4, 8)
    for world in worlds:
        per_rank = []
        pair_counts = []
        for r in range(world):
            lo, hi = D.row_block(r, world, n)
            blk_names = names[lo:hi]
            v = make_vec()
            pb = v.prepare(blk_names)

            def vectorise():
                v2 = make_vec()
                v2.fit_prepared([pb])          # (the sharded fit's table all-reduce is in the collective model below)
                m = v2.transform_prepared(pb)
                m.free()
                return None
            t_vec, k_vec, _ = ms(vectorise)
            plo, phi, pstep = (0, n_index, 1) if world == 1 else D.selfjoin_share(n_index, r, world)

            def pass1():
                got = ctx.selfjoin_range(A, post, 10, 0.8, plo, phi, pstep)
                assert got is not None
                return got
            warm = pass1()                      # (the first call allocates the pair list: not part of a steady step)
            warm[0].free()
            ctx.device_free(warm[1])
            best = None
            for _ in range(2):
                t_p1, k_p1, got = ms(pass1, reps=1)
                if best is None or t_p1 < best[0]:
                    if best is not None:
                        best[2][0].free()
                        ctx.device_free(best[2][1])
                    best = (t_p1, k_p1, got)
                else:
                    got[0].free()
                    ctx.device_free(got[1])
            t_p1, k_p1, got = best
            res, ptr, n_pairs, words = got
            pair_counts.append(n_pairs)
            # the OTHER form of the step (distributed.sharded_topn without the self-join form): the rank's block of rows
            # against all columns, one-sided -- twice the pairs of the self-join form in all, but no pair exchange, no
            # merge, no expansion of groups, no launch over parts

            def row_block_form():
                blk = A.row_block(lo, hi)
                out = ctx.spgemm_topn(blk, post, 10, 0.8, True)
                out.free()
                blk.free()
                return None
            if world > 1:
                row_block_form()
                t_rb, k_rb, _ = ms(row_block_form)
            else:
                t_rb, k_rb = float("nan"), {}
            per_rank.append({"rank": r, "rows": hi - lo, "range": [plo, phi, pstep], "vectorise_ms_wall": t_vec,
                             "vectorise_ms_kernels": k_vec.get("tokenize", 0) + k_vec.get("vocab", 0) + k_vec.get("weight", 0),
                             "pass1_ms_wall": t_p1, "pass1_ms_kernel": k_p1.get("spgemm_topn", 0.0),
                             "pass1_ms_kernel_alone": k_p1.get("spgemm_kernel", 0.0), "pairs": int(n_pairs),
                             "row_block_form_ms_wall": t_rb,
                             "_res": res, "_ptr": ptr, "_words": words})
        # the merge needs ALL ranks' pairs: concatenate them on the device (what the all-gather delivers)
        import torch
        all_pairs = []
        for pr in per_rank:
            if pr["pairs"]:
                t = torch.as_tensor(D.DeviceTensorView(pr["_ptr"], pr["pairs"] * pr["_words"], "<i4"), device="cuda")
                all_pairs.append(t.clone())
        pairs_all = torch.cat(all_pairs) if all_pairs else torch.zeros(0, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        for pr in per_rank:
            plo, phi, pstep = pr["range"]
            words = pr["_words"]

            def merge():
                # the driver's call: merge of the pairs that point into the range + (index over groups) expansion of the
                # range's groups into their member rows
                return ops.selfjoin_merge({"res": pr["_res"], "ptr": pr["_ptr"], "n": pr["pairs"], "words": words, "post": post},
                                          pairs_all, plo, phi, pstep)
            t_m, k_m, blk = ms(merge, reps=1)
            pr["merge_ms_wall"] = t_m
            pr["rows_out"] = int(blk.dims()[0])
            blk.free()
            for k in ("_res", "_ptr", "_words"):
                pr.pop(k)
        # collectives (model): df table all-reduce (2 x table x (N-1)/N through the ring), CSR all-gather (a rank receives
        # the other ranks' blocks), pair all-gather
        table = 4 * (1 << 21)
        csr_bytes = nnz * (4 + s) + 4 * n
        pair_bytes = sum(pair_counts) * (12 if s == 4 else 16)
        frac = (world - 1) / world
        coll_ms = 0.0 if world == 1 else (2 * table * frac + csr_bytes * frac + pair_bytes * frac) / (LINK_GBPS * 1e6) + 3 * COLLECTIVE_LATENCY_MS
        crit = max(p["vectorise_ms_wall"] + t_post + p["pass1_ms_wall"] + p["merge_ms_wall"] for p in per_rank) + coll_ms
        crit_kernels = max(p["vectorise_ms_kernels"] + k_post.get("postings", 0) + p["pass1_ms_kernel"] + p["merge_ms_wall"]
                           for p in per_rank) + coll_ms
        # ... and the critical path of the row-block form: no pairs to exchange, nothing to merge
        coll_rb = 0.0 if world == 1 else (2 * table * frac + csr_bytes * frac) / (LINK_GBPS * 1e6) + 2 * COLLECTIVE_LATENCY_MS
        crit_rb = (max(p["vectorise_ms_wall"] + t_post + p["row_block_form_ms_wall"] for p in per_rank) + coll_rb) if world > 1 else float("nan")
        if world == 1 or one_gpu is None:
            one_gpu = crit
        slow = max(per_rank, key=lambda p: p["pass1_ms_wall"])
        report["ranks"][str(world)] = {"critical_path_ms": crit, "critical_path_ms_kernels_only": crit_kernels,
                                       "speedup_vs_1": one_gpu / crit, "collectives_ms_model": coll_ms,
                                       "slowest_pass1_ms": slow["pass1_ms_wall"], "fixed_ms": max(p["vectorise_ms_wall"] for p in per_rank) + t_post,
                                       "merge_ms": max(p["merge_ms_wall"] for p in per_rank),
                                       "row_block_form_critical_path_ms": crit_rb, "per_rank": per_rank}
        print(f"N={world}: critical path {crit:7.3f} ms (kernels only {crit_kernels:7.3f})  = vectorise "
              f"{max(p['vectorise_ms_wall'] for p in per_rank):.3f} + index {t_post:.3f} + slowest range {slow['pass1_ms_wall']:.3f} "
              f"+ merge {max(p['merge_ms_wall'] for p in per_rank):.3f} + collectives (model) {coll_ms:.3f}   "
              f"speed-up {one_gpu / crit:.2f}x;  ranges' pass 1: {[round(p['pass1_ms_wall'], 2) for p in per_rank]}"
              f" (kernel alone: {[round(p['pass1_ms_kernel_alone'], 2) for p in per_rank]})")
        if world > 1:
            print(f"      row-block form instead: critical path {crit_rb:7.3f} ms ({one_gpu / crit_rb:.2f}x); the ranks' one-sided multiplies "
                  f"{[round(p['row_block_form_ms_wall'], 2) for p in per_rank]}")
    print("JSON " + json.dumps(report))


if __name__ == "__main__":
    main()
