#!/usr/bin/env python
"""bench.py -- hot-path throughput of string_grouper_amd on MI355X.

One "step" = one pass of the hot path over one batch of synthetic input that is already resident
in HBM: n-gram tokenise + vocabulary/df (K1), tf-idf weight + L2 normalise (K2), inverted index (K3),
thresholded sparse top-n multiply (K4p / K4).  Workload at N=1: the configuration BASELINE.json's metric
is quoted on -- 663k-name self-join, 3-grams, ntop=10, min_sim=0.8, fp32 -- on SynthNames-v1
(the sec__edgar list is not distributable; see string_grouper_amd/synth.py).

    python bench.py --gpus N --steps K --warmup W

``--gpus N`` with N > 1 and no rank environment re-executes itself under ``torch.distributed.run`` (N ranks,
one per GPU, RCCL); launched BY torch.distributed.run it reads RANK / LOCAL_RANK / WORLD_SIZE.  N > 1 is strong
scaling of the same workload (string_grouper_amd/distributed.py): every rank vectorises its block of the rows (one
all-reduce of the df table, one all-gather of the CSR blocks), builds the index, and scores the pairs (i, j <= i) of its
range of the self-join form (one all-gather of the mirrored pairs, merge); results stay on the ranks.

Prints ONE JSON line (rank 0):
  value / ms_per_step   the hot path, inputs in HBM (the contract's metric)
  roofline              the multiply kernel of the step, live HIP-event time on the library's stream, priced on its
                        own algorithmic bytes
  exact_kernel          K4 timed live on the same input; results compared bit for bit
  end_to_end            wall-clock of the PUBLIC match_strings() incl. host string preparation, PCIe both ways, the
                        device match list (K6) and the pandas frames, fp32 and fp64, with the split -- this is what
                        BASELINE.json's "match_strings wall-clock" target refers to; never `value`
  cpu_baseline          the reference's match_strings call sequence (oracle/ref_pipeline.py: sklearn + the C port of
                        sparse_dot_topn) on 4 host cores in a child process, bounded sample (oracle/baseline.py);
                        all_cores: the multiply leg on every core
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
# `value` is the device hot path with its inputs resident in HBM (the bench contract); BASELINE.json's "match_strings
# rows/sec" -- the public API end to end, host preparation, PCIe and pandas frames included -- is the SAME line's
# `match_strings_rows_per_s` (the name must not suggest that `value` is the latter)
METRIC = ("hot-path rows/sec (the device path of match_strings: tokenise + tf-idf + postings + SpGEMM-topn, inputs resident "
          "in HBM), 663k-name self-join ntop=10 min_sim=0.8; match_strings end to end: match_strings_rows_per_s")


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="timed steps; 0 (default): as many as fill a timed region of >= 2 s")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--rows", type=int, default=663000, help="names in the self-join (metric config: 663000)")
    ap.add_argument("--top-n", type=int, default=10)
    ap.add_argument("--min-similarity", type=float, default=0.8)
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--cpu-sample-rows", type=int, default=0, help="left rows of the CPU baseline (0 = auto)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-exact-kernel", action="store_true", help="skip the live run of the exact kernel K4")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the match_strings() wall-clock runs (fp32 + fp64)")
    ap.add_argument("--end-to-end", action="store_true", help=argparse.SUPPRESS)      # round-1 flag: now the default
    ap.add_argument("--no-side-runs", action="store_true", help="skip the SG_COLLAPSE=0 and other-dtype runs of the step")
    ap.add_argument("--no-config3", action="store_true", help="N > 1: skip the nested 5 M-name block (configs3_5M)")
    ap.add_argument("--config3-rows", type=int, default=5_000_000, help=argparse.SUPPRESS)
    ap.add_argument("--dup-frac", type=float, default=None,
                    help="share of names whose TF-IDF row repeats an earlier name's (SynthNames-v1 as surveyed: 0.165; the "
                         "reference's README: 0.0026); given: string_grouper_amd.synth.synth_names(dup_frac=...) is the workload")
    ap.add_argument("--cpu-cores", type=int, default=4, help="cores of the reference CPU leg (README: 4)")
    ap.add_argument("--cpu-full", action="store_true", help=argparse.SUPPRESS)        # round-3 flag: now the default
    ap.add_argument("--cpu-sample", action="store_true", help="CPU baseline: bounded legs only (a 40 000-name run of the "
                                                              "reference + ~10 s of the multiply), composed and scaled -- "
                                                              "instead of the default, the unmodified reference's "
                                                              "match_strings on ALL rows (~95 s on 4 cores at 663 k)")
    return ap.parse_args()


def respawn_under_torchrun(args) -> None:
    """``python bench.py --gpus N`` (N > 1) outside a launcher: become ``torch.distributed.run`` with N ranks."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.execv(sys.executable, cmd)


def main():
    args = parse_args()
    args.cpu_full = not args.cpu_sample
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        respawn_under_torchrun(args)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        # N > 1 hardening: what a rank writes to stderr is also kept per rank (the launcher interleaves the ranks'
        # streams), a hung collective ends with an exception after three minutes instead of hanging the job, and rank 0
        # prints ONE JSON line whatever happens -- with "error" instead of a value when a rank has died.
        import faulthandler
        log_dir = os.environ.get("SG_BENCH_LOG_DIR", "/tmp")
        try:
            rank_log = open(os.path.join(log_dir, f"sg_bench_rank{rank}.log"), "w")
            faulthandler.enable(rank_log)
            faulthandler.dump_traceback_later(900, file=rank_log)      # where a rank sits if the job takes this long
        except OSError:
            rank_log = None
        try:
            run(args)
        except BaseException as e:  # noqa: BLE001 -- reported, then re-raised
            import traceback
            if rank_log is not None:
                traceback.print_exc(file=rank_log)
                rank_log.flush()
            if rank == 0 and not isinstance(e, SystemExit):
                sys.stdout.flush()
                os.write(_JSON_FD[0] if _JSON_FD else 1, (json.dumps({
                    "metric": METRIC,
                    "value": None, "unit": "rows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                    "higher_is_better": True, "scaling": "strong", "dtype": args.dtype,
                    "error": f"{type(e).__name__}: {e}"[:500],
                    "rank_logs": os.path.join(log_dir, "sg_bench_rank<r>.log")}) + "\n").encode())
            raise
        return
    run(args)


_JSON_FD = []      # the descriptor the one JSON line goes to (stdout as it was when the process started)


def run(args):
    # RCCL prints a version banner to stdout when the first communicator is created; the contract is ONE
    # JSON line on stdout, so everything else this process (and its libraries) writes goes to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    _JSON_FD.append(json_fd)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1 or os.environ.get("SG_BENCH_FORCE_DIST") == "1"   # (the latter: 1-rank RCCL self-test)
    dtype = np.float32 if args.dtype == "f32" else np.float64

    import torch
    # SG_BENCH_BACKEND=gloo: the N-rank step on a box with fewer GPUs than ranks (the builder's one-GPU box): all ranks on the
    # devices there are, collectives staged through the host (string_grouper_amd/distributed.py: transport) -- the device ops
    # of every rank are the real ones, the transport is not RCCL and the ranks share a GPU, so the line says "backend": "gloo"
    # and its value is no scaling figure.
    backend = os.environ.get("SG_BENCH_BACKEND", "nccl")
    if backend == "gloo":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        import datetime
        limit = datetime.timedelta(seconds=int(os.environ.get("SG_BENCH_COLLECTIVE_TIMEOUT", "180")))
        if backend == "gloo":
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=limit)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank), timeout=limit)
    from string_grouper_amd import _native as N
    from string_grouper_amd.synth import synth_names
    from string_grouper_amd.vectorizer import HipTfidfVectorizer

    stream = torch.cuda.current_stream().cuda_stream
    ctx = N.Context(local_rank, stream=stream if stream else None)

    names = synth_names(args.rows, 1234, dup_frac=args.dup_frac)
    make_vec = lambda: HipTfidfVectorizer(dtype=dtype, ctx=ctx)  # noqa: E731
    dist_mode = os.environ.get("SG_BENCH_DIST_MODE", "sharded")
    whole_column_here = (not distributed) or (rank == 0 and dist_mode != "sharded")
    prepared = make_vec().prepare(names) if whole_column_here else None   # strings -> HBM (untimed)

    # N > 1: every rank holds ITS block of the string column in HBM before the timed region starts (the list is
    # generated identically on every rank; only the block is uploaded).  SG_BENCH_DIST_MODE=strings|csr selects
    # round 1's forms (whole column broadcast from rank 0 and vectorised everywhere / CSR broadcast).
    dev_strings = (None, None)
    local_block = ops = None
    if distributed:
        from string_grouper_amd import distributed as D
        if dist_mode == "sharded":
            lo, hi = D.row_block(rank, world, args.rows)
            local_block = make_vec().prepare(names[lo:hi])
            ops = D.HipOps(ctx, make_vec)
        elif rank == 0:
            dev_strings = D.strings_to_device_tensors(prepared, torch.device("cuda", local_rank))   # HBM-resident input

    def step():
        if distributed:
            if dist_mode == "sharded":
                # tokenise the local block, all-reduce the document frequencies, weight the local block, all-gather
                # the CSR blocks, build the inverted index, multiply the local rows; results stay on the rank
                res, _ = D.distributed_self_join(ops, local_block, args.top_n, args.min_similarity)
                ctx.sync()
                return res
            if dist_mode == "csr":
                res, _, _ = D.sharded_self_join(ctx, prepared, make_vec, args.top_n, args.min_similarity)
                return res
            local, _, _ = D.broadcast_strings(ctx, *dev_strings)
            res, _, _ = D.sharded_self_join_replicated(ctx, local, make_vec, args.top_n, args.min_similarity)
            return res
        return one_gpu_step(make_vec)

    def one_gpu_step(factory, prep=None, top_n=None, thr=None):
        prep = prepared if prep is None else prep
        vec = factory()
        vec.fit_prepared([prep])
        A = vec.transform_prepared(prep)
        post = ctx.postings_build(A)
        res = ctx.spgemm_topn(A, post, args.top_n if top_n is None else top_n, args.min_similarity if thr is None else thr, True)
        ctx.sync()
        res._keep = (A, post, vec)
        return res

    def side_run(factory, k=12, prep=None, top_n=None, thr=None):
        """The same step under another setting (other dtype, a switch of the context, another list, other parameters of the
        public API), timed like the main region on a shorter one: (ms per step, the dominant kernel's ms, its stats)."""
        one_gpu_step(factory, prep, top_n, thr).free()
        one_gpu_step(factory, prep, top_n, thr).free()
        torch.cuda.synchronize()
        t = time.perf_counter()
        kms, st = [], None
        for _ in range(k):
            r = one_gpu_step(factory, prep, top_n, thr)
            st = ctx.stats()
            kms.append(st["ms_spgemm_kernel"] or st["ms_spgemm_topn"])
            r.free()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / k * 1e3, float(np.mean(kms)), st

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        r = step()
        r.free()
    if args.steps <= 0:
        # default: a timed region of at least two seconds (0.3 s of a 27 s run is thin) -- one more untimed
        # step is timed to size it; every rank arrives at the same count (max over ranks)
        barrier()
        tc = time.perf_counter()
        r = step()
        r.free()
        barrier()
        one = time.perf_counter() - tc
        if distributed:
            tt = torch.tensor([one], dtype=torch.float64, device="cuda")
            D._all_reduce(tt, dist.ReduceOp.MAX)
            one = float(tt.item())
        args.steps = int(max(5, min(2000, np.ceil(2.0 / max(one, 1e-4)))))
    barrier()
    if distributed:
        D.reset_collective_tally()
    t0 = time.perf_counter()
    k4_ms = []
    stats = None
    last = None
    for _ in range(args.steps):
        if last is not None:
            last.free()
        last = step()
        stats = ctx.stats()
        k4_ms.append(stats["ms_spgemm_kernel"] or stats["ms_spgemm_topn"])      # the dominant kernel alone
        out_nnz = stats["out_nnz"]
    barrier()
    elapsed = time.perf_counter() - t0
    coll_tally = {k: list(v) for k, v in D.COLLECTIVE_TALLY.items()} if distributed else None   # (the timed steps' own)
    index_rows = index_bytes = None
    if not distributed:
        index_rows = int(ctx.postings_rows(last._keep[1])[0])    # < rows: identical rows indexed once (include/sg_hip.h)
        index_bytes = ctx.postings_bytes(last._keep[1])          # what the pruned multiply reads of the index while it runs
    if distributed and dist_mode == "sharded":
        # (the multi-GPU self-join form merges the mirrored pairs after the multiply's own count: count the rank's rows)
        out_nnz = int(ops.topn_tensors(last)[2].sum().item())
    last.free()
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        D._all_reduce(t, dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # per-rank K4 work: MACs and bytes are per row block; sum over ranks for the job totals
        w = torch.tensor([stats["macs"], stats["spgemm_bytes"], out_nnz], dtype=torch.float64, device="cuda")
        D._all_reduce(w, dist.ReduceOp.SUM)
        job_macs, job_bytes, job_nnz = (float(x) for x in w.tolist())
    else:
        job_macs, job_bytes, job_nnz = float(stats["macs"]), float(stats["spgemm_bytes"]), float(out_nnz)

    # N > 1: BASELINE.json names the 5 M self-join (configs[3]) as THE 8-GPU configuration; `value` stays the 663 k job of the
    # metric, and the same step on 5 M names is timed beside it (every rank: its block of the column in HBM, three timed
    # steps bracketed like the main region, max over ranks) -- a nested block of the one JSON line.
    config3 = None
    if distributed and dist_mode == "sharded" and not args.no_config3 and args.rows < args.config3_rows:
        last = None
        try:
            n3 = args.config3_rows
            names3 = synth_names(n3, 1234)
            lo3, hi3 = D.row_block(rank, world, n3)
            block3 = make_vec().prepare(names3[lo3:hi3])
            del names3
            step3 = lambda: D.distributed_self_join(ops, block3, args.top_n, args.min_similarity)[0]   # noqa: E731
            for _ in range(2):
                step3().free()
            ctx.sync()
            barrier()
            D.reset_collective_tally()
            t3 = time.perf_counter()
            k3 = 3
            nnz3 = 0
            for _ in range(k3):
                r3 = step3()
                ctx.sync()
                nnz3 = int(ops.topn_tensors(r3)[2].sum().item())
                r3.free()
            barrier()
            tally3 = {k: [v[0] / k3, v[1] / k3] for k, v in D.COLLECTIVE_TALLY.items()}    # (the steps' own, before the timing's)
            e3 = torch.tensor([time.perf_counter() - t3, float(nnz3)], dtype=torch.float64, device="cuda")
            tmax = e3[:1].clone()
            D._all_reduce(tmax, dist.ReduceOp.MAX)
            tsum = e3[1:].clone()
            D._all_reduce(tsum, dist.ReduceOp.SUM)
            sec3 = float(tmax.item()) / k3
            config3 = {"workload": f"{n3}-name self-join" + (" (BASELINE.json configs[3])" if n3 == 5_000_000 else " (configs[3] at a reduced size: --config3-rows)") + ", same step, same ranks", "rows": n3, "steps": k3,
                       "ms_per_step": sec3 * 1e3, "rows_per_s": n3 / sec3, "matches": int(tsum.item()),
                       "collectives_per_step": tally3,
                       "form": ("self-join form over interleaved shares" if D.selfjoin_form_wanted(n3, world) else "row blocks")}
            del block3
            ctx.trim()
        except Exception as e:       # the nested block must not lose the line of the metric
            config3 = {"error": repr(e)[:300]}

    if rank != 0:
        if distributed:
            dist.destroy_process_group()
        return

    ms_per_step = elapsed / args.steps * 1e3
    k4_avg_ms = float(np.mean(k4_ms))
    pruned = stats["prune_rows"] > 0
    symmetric = bool(stats.get("prune_symmetric"))
    # dominant kernel of the step.  Pruned multiply: ITS OWN algorithmic bytes (4 B per filter posting it
    # streams + one packed row of B per pair it scores exactly + A + out), not the bytes of the products it
    # proved unnecessary.  Exact multiply: the stream model, (4+s) B per intermediate product + A + out.
    k4_bytes = stats["prune_bytes"] if pruned else stats["spgemm_bytes"]
    achieved = k4_bytes / (k4_avg_ms * 1e-3) / 1e9        # this rank's launch group, GB/s
    result = {
        "metric": METRIC,
        "value": args.rows / (elapsed / args.steps),
        "unit": "rows/s",
        # `value` counts the CALLER's rows; identical strings give identical rows and the library indexes (and multiplies)
        # one per group, then expands the result to all rows (sg_collapse.hip) -- the line says how many that were
        "rows_indexed": index_rows, "rows_indexed_per_row": (index_rows / args.rows) if index_rows else None,
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": args.dtype,
        "backend": (backend if distributed else None),
        # what rank 0's collectives moved per step: {kind: [calls, bytes received]} (string_grouper_amd/distributed.py)
        "collectives_per_step": ({k: [v[0] / args.steps, v[1] / args.steps] for k, v in coll_tally.items()}
                                 if distributed else None),
        "data": "synthetic (SynthNames-v1 seed 1234; sec__edgar names are not distributable)" +
                ("" if args.dup_frac is None else f"; repeats thinned to {args.dup_frac:g} of the names (synth_names(dup_frac=...))"),
        "config": {"workload": f"{args.rows}-name self-join (BASELINE.json configs[2] on the synthetic stand-in)",
                   "ngram_size": 3, "max_n_matches": args.top_n, "min_similarity": args.min_similarity,
                   "parallelism": "single GPU" if world == 1 else
                   f"{world} GPUs, one process each: rows of the string column in {world} contiguous blocks; tokenise local "
                   f"block -> all-reduce df table -> weight local block -> all-gather CSR -> inverted index -> multiply: "
                   + ("self-join form over interleaved shares of the rows (pairs j <= i, one all-gather of the mirrored pairs, merge)"
                      if (distributed and dist_mode == "sharded" and D.selfjoin_form_wanted(args.rows, world))
                      else "the local rows against all columns, no collective in the multiply") + f" ({dist_mode})"},
        # launch groups of one step (HIP events on the library's stream); spgemm_topn = the multiply's whole group (pruned
        # kernel + pair-list pass in the self-join form), of which the dominant kernel alone is roofline.avg_ms
        "kernels_ms": {k[3:]: round(v, 4) for k, v in stats.items() if k.startswith("ms_") and k != "ms_spgemm_kernel"},
        # SG_* switches in force (read from the environment once, when the context was created; include/sg_hip.h)
        "options": ctx.options(),
        "matches": int(job_nnz),
        "macs": int(job_macs),
        "configs3_5M": config3,
        # `bound` names the roofline the fraction is priced against (the contract knows "hbm" and "mfma"; no dense
        # contraction here).  `limited_by` says what the counters say of the kernel (profiles/): it
        # waits for dependent misses at the occupancy its LDS tile allows; `l3_resident`: the index it reads fits the
        # 256 MiB Infinity Cache, i.e. most of `traffic` (FETCH_SIZE counts Infinity-Cache hits) never reaches HBM.
        "roofline": {"bound": "hbm",
                     "limited_by": ("memory latency at 16 single-wave workgroups per CU (waves wait for L2 misses about half of "
                                    "their cycles; VALU issue in valu_issue_frac), not HBM bandwidth"),
                     # (what the kernel READS of the index -- the 8-bit records at the units their rows reach, not the 256 B a
                     #  record is allocated at; sg_postings_bytes)
                     "l3_resident": (bool(index_bytes <= 256 * 1024 * 1024) if index_bytes else None),
                     "hbm_share": (("unmeasured: the index the kernel reads fits the Infinity Cache, `traffic` is what crosses "
                                    "the L2's memory side and counts Infinity-Cache hits")
                                   if index_bytes and index_bytes <= 256 * 1024 * 1024 else
                                   ("unmeasured: no counter separates Infinity-Cache hits from HBM reads; the index is larger than "
                                    "the cache, `traffic` is an upper bound of the HBM bytes")),
                     "index_bytes_read_by_the_kernel": index_bytes,
                     "kernel": ("spgemm_topn_pruned_kernel<SYM> + pair-list pass (K4p, self-join form)" if symmetric else
                                "spgemm_topn_pruned_kernel (K4p)") if pruned else "spgemm_topn_kernel (K4)",
                     "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                     "traffic": None, "algorithmic_bytes_per_launch": int(k4_bytes), "avg_ms": k4_avg_ms,
                     "note": ("algorithmic = 4 B per filter posting streamed + (16 + 4 x mean row entries) B of the 8-bit copy per "
                              "candidate of the first filter + one packed row per pair scored exactly + A + out")
                     if pruned else
                     "algorithmic = stream model (4+s) B per intermediate product + A + out"},
    }
    if pruned:
        result["pruning"] = {"rows": stats["prune_rows"], "postings_streamed": stats["prune_postings"],
                             "of_intermediate_products": stats["macs"],
                             # candidates the first filter records -> pairs the second filter (8-bit copy of the candidate's
                             # row) lets through to the exact scoring
                             "candidates_of_the_first_filter": stats["prune_survivors"],
                             "pairs_scored_exactly": stats["prune_scored"],
                             "rows_handed_to_exact_kernel": stats["exact_rows"], "self_join_form": symmetric}
        if index_rows is not None and index_rows != args.rows:
            # identical strings give identical rows: the index holds one representative per group, the multiply runs on
            # the groups and its result is expanded to all rows (sg_collapse.hip; SG_COLLAPSE=0 switches it off).  Exact:
            # the matches of all `rows` rows are the reference's, bit for bit.
            result["pruning"]["identical_rows_indexed_once"] = {"rows": args.rows, "index_rows": index_rows}

    # measured memory-side traffic and VALU issue fraction of the same kernel on the same workload, from the committed PMC
    # passes -- quoted only while the kernel sources are the ones the passes ran on (string_grouper_amd/_provenance.py:
    # `source_sha`; a later edit of the kernels turns the fields into "stale" instead of carrying old counters along)
    if world == 1:
        from string_grouper_amd._provenance import committed_counters
        result["roofline"].update(committed_counters(ROOT, args.rows, args.dtype,
                                                     ("K4p-sym" if symmetric else "K4p") if pruned else "K4", k4_avg_ms))

    # (the public API is timed right behind the main region, before the side runs and the exact kernel.  Its figure varies
    #  between runs on the shared boxes: the download of the match list -- 22 MB, device to host -- takes 2 ms on most runs and
    #  20 - 30 on some, whatever the destination and the copy engine, and the end-to-end figure then reads 0.045 s instead of
    #  0.022; `match_list_and_download_s` in the split tells which it was: scripts/e2e_variance.sh)
    if world == 1 and not args.no_end_to_end:
        # the public API end to end: pandas Series in, match frame out (host preparation, PCIe, K1-K4p, K6, frames)
        import pandas as pd
        import string_grouper_amd as sga
        import string_grouper_amd.engine as E
        eng = E.HipEngine(ctx)
        E.set_engine(eng)
        series = pd.Series(names)
        # Device-to-host bandwidth as the API call will find it.  On some runs on the shared boxes the match list's 22 MB come
        # down at ~1 GB/s instead of 10-50 (scripts/e2e_ab.sh, e2e_variance.sh; no cause found that the library controls) and the
        # figure below records the box, not the library.  Probe until the link is up to speed, five seconds at most; what was
        # seen is in the line (`d2h_probe_gbps`), and `match_list_and_download_s` of the split says how the call itself fared.
        d2h_seen = []
        try:
            dev_buf = torch.empty(32 << 20, dtype=torch.uint8, device="cuda")
            host_buf = torch.empty(32 << 20, dtype=torch.uint8).pin_memory()
            t_probe = time.perf_counter()
            while time.perf_counter() - t_probe < 5.0:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                host_buf.copy_(dev_buf)
                torch.cuda.synchronize()
                d2h_seen.append(round((32 << 20) / (time.perf_counter() - t0) / 1e9, 2))
                if len(d2h_seen) >= 3 and min(d2h_seen[-3:]) >= 10.0:
                    break
            del dev_buf, host_buf
        except Exception as e:  # noqa: BLE001 -- a probe, not a requirement
            d2h_seen.append(f"probe failed: {type(e).__name__}")
        result["end_to_end"] = {"what": "wall-clock of string_grouper_amd.match_strings(pd.Series) -> DataFrame, best of 3 "
                                        "after warm-up calls (until a call is no faster than the one before, at most 8); includes host string preparation, H2D, the device hot "
                                        "path, the device match list (K6), D2H and the pandas frames",
                                "d2h_probe_gbps": d2h_seen[:3] + (["..."] if len(d2h_seen) > 6 else []) + d2h_seen[3:][-3:]}
        for dname, dt in (("f32", np.float32), ("f64", np.float64)):
            best, split, n_match = None, None, 0
            # warm-up: until a call is no longer faster than the one before it (at most eight)
            warm, prev = 0, None
            while warm < 8:
                t0 = time.perf_counter()
                df = sga.match_strings(series, max_n_matches=args.top_n, min_similarity=args.min_similarity, tfidf_matrix_dtype=dt)
                t = time.perf_counter() - t0
                del df
                warm += 1
                if prev is not None and t > 0.9 * prev:
                    break
                prev = t
            for rep in range(3):
                t0 = time.perf_counter()
                df = sga.match_strings(series, max_n_matches=args.top_n, min_similarity=args.min_similarity,
                                       tfidf_matrix_dtype=dt)
                t = time.perf_counter() - t0
                n_match = len(df)
                if best is None or t < best:
                    best = t
                    split = dict(eng.timings)
                del df
            device = split.get("vectorise_s", 0.0) + split.get("multiply_s", 0.0) + split.get("match_list_and_download_s", 0.0)
            split["validation_and_frames_s"] = best - device - split.get("prepare_and_upload_s", 0.0)
            result["end_to_end"][dname] = {"seconds": best, "rows_per_s": args.rows / best, "match_rows": n_match, "warm_up_calls": warm,
                                           "split": {k: round(v, 5) for k, v in split.items()}}

    if world == 1 and not args.no_side_runs:
        # the step WITHOUT the collapse of identical rows (every one of the 663 000 rows indexed and multiplied: 16.5 % of
        # SynthNames-v1's names repeat, a property of the generator, not of sec__edgar) ...
        ctx.set_option("SG_COLLAPSE", "0")
        ms, kms, st_nc = side_run(make_vec)
        ctx.set_option("SG_COLLAPSE", None)
        nc_bytes = st_nc["prune_bytes"] if st_nc["prune_rows"] > 0 else st_nc["spgemm_bytes"]
        # (first-class since round 6: what carries over to a list that does not repeat itself)
        result["value_without_identical_rows"] = args.rows / (ms * 1e-3)
        result["without_row_collapse"] = {"ms_per_step": ms, "rows_per_s": args.rows / (ms * 1e-3), "kernel_ms": kms,
                                          "kernel_frac_of_hbm_peak": nc_bytes / (kms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                          "steps": 12, "option": "SG_COLLAPSE=0"}
        # ... on the list with the repeats of a REAL company list: 0.3 % of the names (the reference's README.md:80-95 counts
        # 1 747 names in groups of identical names among 663 000), the library's defaults otherwise
        if args.dup_frac is None:
            names_lo = synth_names(args.rows, 1234, dup_frac=0.003)
            prep_lo = make_vec().prepare(names_lo)
            del names_lo
            ms, kms, st_lo = side_run(make_vec, prep=prep_lo)
            lo_bytes = st_lo["prune_bytes"] if st_lo["prune_rows"] > 0 else st_lo["spgemm_bytes"]
            result["low_duplicates"] = {"dup_frac": 0.003, "ms_per_step": ms, "rows_per_s": args.rows / (ms * 1e-3), "kernel_ms": kms,
                                        "kernel_frac_of_hbm_peak": lo_bytes / (kms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "steps": 12,
                                        "matches": st_lo["out_nnz"], "rows_indexed": st_lo["prune_rows"] + 1,
                                        "workload": "synth_names(rows, 1234, dup_frac=0.003): every repeat of SynthNames-v1 beyond "
                                                    "0.3 % of the names replaced by a fresh name"}
            del prep_lo
        # ... and in the other value type (fp64 is the reference's DEFAULT tfidf_matrix_dtype, string_grouper.py:18)
        other = "f64" if args.dtype == "f32" else "f32"
        odt = np.float64 if other == "f64" else np.float32
        ms, kms, st_o = side_run(lambda: HipTfidfVectorizer(dtype=odt, ctx=ctx))
        o_bytes = st_o["prune_bytes"] if st_o["prune_rows"] > 0 else st_o["spgemm_bytes"]
        result[other] = {"ms_per_step": ms, "rows_per_s": args.rows / (ms * 1e-3), "kernel_ms": kms,
                         "kernel_frac_of_hbm_peak": o_bytes / (kms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "steps": 12,
                         "matches": st_o["out_nnz"]}

    if world == 1 and not args.no_side_runs:
        # ... and at other values of the two parameters the reference leaves to the caller (string_grouper.py:192-193): the whole
        # step again -- every one rebuilds its indexes, the second one of a form included -- on the same list.  Which form of
        # the multiply the library picked is read off its statistics ("Which form runs").
        other_settings = []
        for top_n, thr in ((100, 0.8), (10, 0.6), (20, 0.5), (10, 0.35)):
            ms, kms, st_s = side_run(make_vec, k=4, top_n=top_n, thr=thr)
            if st_s["prune_rows"] > 0:
                form = "pruned multiply, " + ("stream form + second filter" if thr >= 0.65 else "tile-by-tile form")
            else:
                form = "exact kernel"
            form += ", self-join form" if st_s["prune_symmetric"] else ", one-sided"
            other_settings.append({"top_n": top_n, "min_similarity": thr, "ms_per_step": ms, "rows_per_s": args.rows / (ms * 1e-3),
                                   "multiply_ms": st_s["ms_spgemm_topn"], "matches": st_s["out_nnz"], "form": form,
                                   "rows_handed_to_exact_kernel": st_s["exact_rows"] if st_s["prune_rows"] > 0 else None, "steps": 4})
        result["other_settings"] = other_settings

    if world == 1 and pruned and not args.no_exact_kernel:
        # the exact kernel (K4) on the same input, timed live beside the pruned one: it is what runs when the
        # data is not cosine-like or top_n > 64, and the pruned result must equal it bit for bit
        ctx.set_option("SG_PRUNE", "0")          # (an option of this context, set explicitly -- the environment is not touched)
        vec = make_vec()
        vec.fit_prepared([prepared])
        A = vec.transform_prepared(prepared)
        post = ctx.postings_build(A)
        ex_ms = []
        for _ in range(2):
            r_ex = ctx.spgemm_topn(A, post, args.top_n, args.min_similarity, True)
            ctx.sync()
            st_ex = ctx.stats()
            ex_ms.append(st_ex["ms_spgemm_topn"])
            if len(ex_ms) < 2:
                r_ex.free()
        ctx.set_option("SG_PRUNE", None)
        r_pr = step()
        h_ex, h_pr = r_ex.to_host(), r_pr.to_host()
        mask = np.arange(h_ex[0].shape[1])[None, :] < h_ex[2][:, None]
        identical = bool(np.array_equal(h_ex[2], h_pr[2]) and np.array_equal(h_ex[0][mask], h_pr[0][mask])
                         and np.array_equal(h_ex[1][mask], h_pr[1][mask]))
        r_ex.free()
        r_pr.free()
        post.free()
        A.free()
        ex_gbps = st_ex["spgemm_bytes"] / (min(ex_ms) * 1e-3) / 1e9
        result["exact_kernel"] = {"kernel": "spgemm_topn_kernel (K4)", "ms": min(ex_ms), "achieved": ex_gbps,
                                  "unit": "GB/s", "frac": ex_gbps / HBM_PEAK_GBPS,
                                  "algorithmic_bytes_per_launch": int(st_ex["spgemm_bytes"]),
                                  "pruned_result_identical": identical,
                                  "note": "stream model (4+s) B per intermediate product + A + out"}

    if world == 1 and not args.no_cpu_baseline:
        # the reference's CPU path on this box's host cores, in a child process pinned to --cpu-cores CPUs
        # (oracle/baseline.py).  The child gets the full TF-IDF matrix from this process (built by the GPU path,
        # parity-checked by the tests) so that its budget goes into what it measures.
        import subprocess
        import tempfile
        from oracle import port as P
        vec = make_vec()
        vec.fit_prepared([prepared])
        A_dev = vec.transform_prepared(prepared)
        A_host = A_dev.to_scipy()
        A_dev.free()
        tmp = tempfile.mkdtemp(prefix="sg_bench_")
        mpath = os.path.join(tmp, "A.npz")
        np.savez(mpath, indptr=A_host.indptr.astype(np.int64), indices=A_host.indices, data=A_host.data,
                 shape=np.array(A_host.shape))
        common = [sys.executable, "-m", "oracle.baseline", "--matrix", mpath, "--rows", str(args.rows), "--top-n",
                  str(args.top_n), "--min-similarity", str(args.min_similarity), "--dtype", args.dtype,
                  "--matches-full", str(result.get("end_to_end", {}).get(args.dtype, {}).get("match_rows", 0))]
        all_cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        try:
            # Default: the UNMODIFIED reference's match_strings on ALL rows (oracle/baseline.py --reference-full: the package
            # travels as oracle/_ref/reference_pkg.zip, its absent sparse_dot_topn wheel = the C port; ~95 s on 4 cores at
            # 663 k) -- measured, nothing scaled; its legs on a 40 000-name run and a bounded sample of the multiply beside
            # it.  Without the archive: the restated pipeline with every left row multiplied (round 4's line, kind "port").
            # --cpu-sample: the bounded legs only, composed.
            run = lambda extra, secs, limit: json.loads(subprocess.run(                                  # noqa: E731
                common + ["--cores", str(args.cpu_cores), "--multiply-seconds", secs] + extra, cwd=ROOT, capture_output=True,
                text=True, timeout=limit).stdout.strip().splitlines()[-1])
            base = run(["--reference-full"] if args.cpu_full else [], "10", 1500)
            if args.cpu_full and base.get("kind") == "port":
                base = run([], "100000", 1500)
            rall = subprocess.run(common + ["--cores", str(min(all_cores, 64)), "--multiply-seconds", "6",
                                            "--multiply-only"], cwd=ROOT, capture_output=True, text=True, timeout=600)
            ball = json.loads(rall.stdout.strip().splitlines()[-1])
            vec_s = base["vectorise"]["seconds_full_estimate"]
            tail_s = base["tail"]["seconds_full_estimate"]
            all_total = vec_s + ball["multiply"]["seconds_full_estimate"] + tail_s
            composed = vec_s + base["multiply"]["seconds_full_estimate"] + tail_s
            full = base.get("reference_full_run")
            is_ref = base.get("kind") == "reference+port"
            legs_from = (f"the unmodified reference's match_strings on {base['reference_run']['rows']} names "
                         f"({base['reference_run']['seconds']:.2f} s; frames equal the restated pipeline's: "
                         f"{base['reference_run']['frames_equal_the_restated_pipeline']})" if is_ref else
                         f"the restated pipeline (oracle/ref_pipeline.py) on {base['small_run']['rows']} names "
                         f"({base['small_run']['seconds']:.2f} s)")
            result["cpu_baseline"] = {
                "value": base["value"], "unit": "rows/s", "cores": base["cores"], "kind": base.get("kind", "port"),
                # nothing is scaled when the reference ran on all rows; otherwise the tokenisation passes and the tail are
                # scaled from the small run (`scaled_from_rows`) and the multiply from its sample unless every row was multiplied
                "extrapolated": bool(full is None),
                "cpu_model": base["cpu_model"], "seconds_full_estimate": base["seconds_full_estimate"],
                "sample": ((f"string_grouper.match_strings(names, max_n_matches={args.top_n}, min_similarity={args.min_similarity}, "
                            f"tfidf_matrix_dtype={args.dtype}, number_of_processes={base['cores']}) of the UNMODIFIED reference "
                            f"package (oracle/_ref/reference_pkg.zip, packed from /root/reference by build(); its absent "
                            f"sparse_dot_topn wheel stood in for by oracle/sdtn_port.c) on ALL {full['rows']} names, child "
                            f"process pinned to {base['cores']} cores: {full['seconds']:.1f} s wall clock, measured, nothing scaled")
                           if full is not None else
                           (f"composed: vectorise + tail legs from {legs_from}, scaled by rows / match rows; multiply = the "
                            f"reference's own block split n_blocks={tuple(base['multiply']['n_blocks'])} of the full problem on "
                            f"oracle/sdtn_port.c for the first {base['multiply']['sample_left_rows']} left rows "
                            f"({base['multiply']['seconds_sample']:.2f} s), scan part scaled by exact MAC count; child process "
                            f"pinned to {base['cores']} cores")),
                "measured_full_run": ({"seconds": full["seconds"], "match_rows": full["match_rows"], "legs_seconds": full["legs"]}
                                      if full is not None else None),
                # the same job composed from bounded legs (cross-check of the measured run; the line of --cpu-sample)
                "composed_estimate": {"seconds": composed, "legs_from": legs_from,
                                      "scaled_from_rows": base["vectorise"].get("scaled_from_rows"),
                                      "split_seconds": {"vectorise_3_passes": vec_s,
                                                        "multiply": base["multiply"]["seconds_full_estimate"], "tail": tail_s},
                                      "multiply_sample_left_rows": base["multiply"]["sample_left_rows"]},
                "all_cores": {"cores": ball["cores"], "cpus_in_affinity_mask": ball.get("cpus_in_affinity_mask"),
                              "cgroup_cpu_quota": ball.get("cgroup_cpu_quota"),
                              "multiply_sample_left_rows": ball["multiply"]["sample_left_rows"],
                              "multiply_seconds_sample": ball["multiply"]["seconds_sample"],
                              "multiply_seconds_full_estimate": ball["multiply"]["seconds_full_estimate"],
                              "value": args.rows / all_total, "unit": "rows/s",
                              "note": "composed, with the multiply leg on every core (tokenisation stays single-threaded in the reference)"},
            }
        except Exception as e:       # a baseline failure must not lose the GPU line
            result["cpu_baseline"] = {"error": repr(e)[:300]}
        finally:
            try:
                os.remove(mpath)
                os.rmdir(tmp)
            except OSError:
                pass
        # the first rows of the C port must equal the GPU rows (parity on the bench workload itself; the
        # whole 663k x 663k comparison is tests/test_parity_gpu.py::test_headline_663k_...)
        sample = min(args.rows, 30000)
        C_cpu = P.sp_matmul_topn_port(A_host[:sample], A_host.T, args.top_n, args.min_similarity, True, min(all_cores, 64))
        r = step()
        C_gpu = r.to_scipy()[:sample]
        r.free()
        result["parity_on_sample"] = bool(np.array_equal(C_gpu.indptr, C_cpu.indptr) and
                                          np.array_equal(C_gpu.indices, C_cpu.indices) and
                                          np.array_equal(C_gpu.data, C_cpu.data))
        if "value" in result.get("cpu_baseline", {}) and "end_to_end" in result:
            e2e = result["end_to_end"].get(args.dtype, {})
            if e2e.get("seconds"):
                result["end_to_end"]["speedup_vs_cpu_baseline_4_cores"] = \
                    result["cpu_baseline"]["seconds_full_estimate"] / e2e["seconds"]

    if "end_to_end" in result and args.dtype in result["end_to_end"]:
        # what BASELINE.json calls match_strings: the public API end to end (host preparation, PCIe, frames); `value` is the
        # device hot path with its inputs resident in HBM, as the bench contract defines it
        result["match_strings_rows_per_s"] = result["end_to_end"][args.dtype]["rows_per_s"]
    sys.stdout.flush()
    os.write(json_fd, (json.dumps(result) + "\n").encode())
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
