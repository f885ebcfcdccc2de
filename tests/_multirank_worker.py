"""One rank of the multi-rank GPU tests (tests/test_multirank_gpu.py): several processes on ONE device (cuda:0), process
group on gloo -- RCCL refuses two ranks of a communicator on the same GPU, so the transport is the host-staged one of
``string_grouper_amd.distributed`` -- driving the REAL device ops (``distributed.HipOps`` on libsg_hip.so) of the N > 1 path:
group-position ranges, the merge of another rank's pairs, the expansion of a rank's groups, the scatter of the gathered
blocks.  Every rank compares ITS block of the result, and the gathered whole, with the CPU port's rows that the parent
process wrote to ``workdir`` (string_grouper/string_grouper.py:733-752 is what the ranks replace).

Test infrastructure; imported by the tests only."""
from __future__ import annotations

import os
import traceback

import numpy as np
import scipy.sparse as sp


def _csr_of(cols, vals, counts, n_cols):
    indptr = np.zeros(len(counts) + 1, np.int64)
    np.cumsum(counts, out=indptr[1:])
    mask = np.arange(cols.shape[1])[None, :] < counts[:, None]
    return sp.csr_matrix((vals[mask], cols[mask], indptr), shape=(len(counts), n_cols))


def _same(got, want) -> str:
    """'' when the two CSR matrices are identical bit for bit (row lengths, columns in order, score bits), else what differs."""
    if got.shape != want.shape:
        return f"shape {got.shape} != {want.shape}"
    gl, wl = np.diff(got.indptr), np.diff(want.indptr)
    if not np.array_equal(gl, wl):
        bad = np.flatnonzero(gl != wl)
        return f"{len(bad)} rows differ in length, first row {bad[0]}: {gl[bad[0]]} != {wl[bad[0]]}"
    if not np.array_equal(got.indices, want.indices):
        return f"columns differ at {int(np.flatnonzero(got.indices != want.indices)[0])}"
    if got.data.dtype != want.data.dtype:
        return f"dtype {got.data.dtype} != {want.data.dtype}"
    if not np.array_equal(got.data, want.data):
        return f"scores differ at {int(np.flatnonzero(got.data != want.data)[0])}"
    return ""


def load_expected(workdir, key):
    z = np.load(os.path.join(workdir, key + ".npz"))
    return sp.csr_matrix((z["data"], z["indices"], z["indptr"]), shape=tuple(z["shape"]))


def save_expected(workdir, key, C):
    np.savez(os.path.join(workdir, key + ".npz"), data=C.data, indices=C.indices, indptr=C.indptr.astype(np.int64),
             shape=np.array(C.shape))


def block_of(res, rank, world, n_rows):
    """(row numbers, CSR) of this rank's block of a result of ``distributed_self_join`` / ``distributed_match``."""
    import torch  # noqa: F401
    from string_grouper_amd import distributed as D
    if isinstance(res, D.TopNRows):
        C = res.to_scipy()
        if res.row_ids is not None:
            rows = res.row_ids.cpu().numpy().astype(np.int64)
        elif res.orig_of is not None:
            rows = res.orig_of[res.lo:res.hi].cpu().numpy().astype(np.int64)
        else:
            rows = np.arange(res.lo, res.hi)
        return rows, C
    lo, hi = D.row_block(rank, world, n_rows)
    return np.arange(lo, hi), res.to_scipy()


def worker(rank: int, world: int, port: int, workdir: str, jobs, ret, backend: str = "gloo"):
    """``jobs``: list of dicts (see tests/test_multirank_gpu.py).  ret[rank] = {check name: '' or what went wrong}.
    ``backend``: "gloo" -- all ranks on cuda:0, collectives on a host copy -- or "nccl": one rank per DEVICE (rank r on
    cuda:r), the product's transport (RCCL over xGMI), device tensors straight into the collectives."""
    out = {}
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        os.environ.pop("SG_DIST_SYM", None)
        os.environ.pop("SG_DIST_INTERLEAVE", None)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # (the host driver only supports dmabuf IPC)
        os.environ.setdefault("OMP_NUM_THREADS", "2")                # (eight ranks on a box of sixteen cores: no 8 x 16 threads)
        import datetime
        import torch
        import torch.distributed as dist
        torch.set_num_threads(2)
        device = rank if backend == "nccl" else 0
        torch.cuda.set_device(device)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=240),
                                    device_id=torch.device("cuda", device))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        try:
            _run(rank, world, workdir, jobs, out, device)
        finally:
            dist.destroy_process_group()
    except BaseException:  # noqa: BLE001 -- reported to the parent, which fails the test with it
        out["exception"] = traceback.format_exc()[-3000:]
    ret[rank] = out


def _run(rank, world, workdir, jobs, out, device=0):
    import pandas as pd
    import torch
    import torch.distributed as dist
    import string_grouper_amd as sga
    import string_grouper_amd.engine as E
    from string_grouper_amd import _native as N
    from string_grouper_amd import distributed as D
    from string_grouper_amd.synth import synth_names
    from string_grouper_amd.vectorizer import HipTfidfVectorizer

    ctx = N.Context(device)                 # the rank's own context and stream (gloo: on the shared device; nccl: its own GPU)
    on_dev = torch.device("cuda", device) if dist.get_backend() == "nccl" else torch.device("cpu")
    for job in jobs:
        tag = job["tag"]
        dtype = np.float64 if job.get("dtype") == "f64" else np.float32
        make_vec = lambda: HipTfidfVectorizer(dtype=dtype, ctx=ctx)          # noqa: E731
        ops = D.HipOps(ctx, make_vec)
        ctx.reset_options()
        for k, v in job.get("options", {}).items():
            ctx.set_option(k, v)
        for k in ("SG_DIST_SYM", "SG_DIST_INTERLEAVE"):
            os.environ.pop(k, None)
        for k, v in job.get("env", {}).items():
            os.environ[k] = v
        top_n, thr = job["top_n"], job["thr"]
        if job["kind"] == "selfjoin":
            names = _names_of(job, synth_names)
            want = load_expected(workdir, job["expected"])
            lo, hi = D.row_block(rank, world, len(names))
            block = make_vec().prepare(names[lo:hi])
            res, _ = D.distributed_self_join(ops, block, top_n, thr)
            ctx.sync()
            st = ctx.stats()
            form = "selfjoin" if isinstance(res, D.TopNRows) else "rowblock"
            if job.get("form") and form != job["form"]:
                out[tag + ":form"] = f"took the {form} form, the job expects {job['form']}"
            if job.get("grouped") is not None:
                grouped = isinstance(res, D.TopNRows) and res.row_ids is not None and res.sel is None
                if form == "selfjoin" and grouped != job["grouped"]:
                    out[tag + ":grouped"] = f"index over groups: {grouped}, the job expects {job['grouped']}"
            rows, C = block_of(res, rank, world, len(names))
            out[tag + ":my_rows"] = _same(C, want[rows])
            n_mine = torch.tensor([len(rows)], device=on_dev)
            dist.all_reduce(n_mine)
            out[tag + ":every_row_once"] = "" if int(n_mine) == len(names) else f"{int(n_mine)} rows in all blocks, {len(names)} names"
            cols, vals, counts = D.gather_topn(ops, res)
            out[tag + ":gathered"] = _same(_csr_of(cols, vals, counts, len(names)), want)
            if form == "selfjoin" and not st["prune_symmetric"]:
                out[tag + ":kernel"] = f"the self-join form did not run the self-join kernel: {st}"
            res.free()
        elif job["kind"] == "match":
            master = synth_names(job["n_master"], job["seed"])
            dups = synth_names(job["n_dups"], job["seed"] + 1, perturb_of=master, perturb_frac=0.5)
            want = load_expected(workdir, job["expected"])
            mlo, mhi = D.row_block(rank, world, len(master))
            dlo, dhi = D.row_block(rank, world, len(dups))
            v = make_vec()
            res, _ = D.distributed_match(ops, v.prepare(master[mlo:mhi]), v.prepare(dups[dlo:dhi]), top_n, thr)
            ctx.sync()
            rows, C = block_of(res, rank, world, len(master))
            out[tag + ":my_rows"] = _same(C, want[rows])
            cols, vals, counts = D.gather_topn(ops, res)
            out[tag + ":gathered"] = _same(_csr_of(cols, vals, counts, len(dups)), want)
            res.free()
        elif job["kind"] == "api":
            # the public API under enable_distributed(): every rank runs the same script on the same Series and gets the same
            # frames -- those of the one-GPU engine (pinned on the reference's own output by tests/test_reference_fixtures.py)
            names = _names_of(job, synth_names)
            dups = synth_names(job["n_dups"], job["seed"] + 1, perturb_of=names, perturb_frac=0.5)
            s_m, s_d = pd.Series(names, name="name"), pd.Series(dups, name="dup")

            def calls():
                return [sga.match_strings(s_m, min_similarity=thr, max_n_matches=top_n, tfidf_matrix_dtype=dtype),
                        sga.match_strings(s_m, s_d, min_similarity=0.7, tfidf_matrix_dtype=dtype),
                        pd.DataFrame(sga.group_similar_strings(s_m, min_similarity=thr, tfidf_matrix_dtype=dtype)),
                        pd.DataFrame(sga.match_most_similar(s_m, s_d, min_similarity=0.7, tfidf_matrix_dtype=dtype)),
                        pd.DataFrame(sga.compute_pairwise_similarities(s_m[:len(s_d)].reset_index(drop=True), s_d,
                                                                       tfidf_matrix_dtype=dtype))]
            old = E._engine
            try:
                E.set_engine(E.HipEngine(ctx))
                want_frames = calls()
                eng = E.enable_distributed(ctx=ctx)
                if not isinstance(eng, E.DistributedHipEngine):
                    out[tag + ":engine"] = f"enable_distributed() gave {type(eng).__name__}"
                got_frames = calls()
            finally:
                E.set_engine(old)
            for i, (w, g) in enumerate(zip(want_frames, got_frames)):
                try:
                    pd.testing.assert_frame_equal(w, g)
                    out[f"{tag}:frame{i}"] = ""
                except AssertionError as e:
                    out[f"{tag}:frame{i}"] = str(e)[:400]
        else:
            out[tag] = f"unknown job kind {job['kind']}"
        dist.barrier()
    ctx.reset_options()


def _names_of(job, synth_names):
    if job.get("distinct"):
        # a column with only a handful of distinct strings (fewer groups than top_n)
        base = synth_names(job["distinct"], job["seed"])
        rng = np.random.default_rng(job["seed"])
        return [base[i] for i in rng.integers(0, len(base), job["n"])]
    names = synth_names(job["n"], job["seed"])
    return names + list(job.get("extra", []))
