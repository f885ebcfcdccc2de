"""world_size-2 gloo tests (CPU) of the multi-GPU orchestration: row blocks, the one CSR broadcast,
the count gather, and that block-wise results concatenate to the unsharded result (the oracle stands
in for the device multiply; the sharding / collective code is the product's)."""
import os
import socket

import numpy as np
import pytest
import scipy.sparse as sp
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from string_grouper_amd import distributed as D


def test_row_blocks_cover_everything():
    for n in (0, 1, 7, 8, 663000, 5000001):
        for world in (1, 2, 3, 8):
            blocks = [D.row_block(r, world, n) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in blocks]
            assert max(sizes) - min(sizes) <= 1


def test_weighted_row_blocks_balance_cost():
    rng = np.random.default_rng(0)
    cost = rng.pareto(1.5, 100000) + 1
    cuts = D.weighted_row_blocks(cost, 8)
    assert cuts[0] == 0 and cuts[-1] == len(cost) and (np.diff(cuts) >= 0).all()
    per = [cost[cuts[i]:cuts[i + 1]].sum() for i in range(8)]
    assert max(per) / (sum(per) / 8) < 1.05


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as O
        from oracle import port as P
        from string_grouper_amd.synth import synth_names
        names = synth_names(3000, 42)
        if rank == 0:
            (A,), _, _ = O.tfidf_sklearn(names, [names], dtype=np.float32)
            t = (torch.from_numpy(A.indptr.astype(np.int64)), torch.from_numpy(A.indices.astype(np.int32)),
                 torch.from_numpy(A.data.copy()), A.shape)
        else:
            t = (None, None, None, None)
        ip, ix, d, shape = D.broadcast_csr(*t, src=0, device=torch.device("cpu"))
        B = sp.csr_matrix((d.numpy(), ix.numpy(), ip.numpy()), shape=shape)
        lo, hi = D.row_block(rank, world, shape[0])
        C_local = P.sp_matmul_topn_port(B[lo:hi], B.T, 10, 0.8, True, 2)
        counts = torch.from_numpy(np.diff(C_local.indptr).astype(np.int32))
        all_counts = D.gather_counts(counts, shape[0])
        C_full = P.sp_matmul_topn_port(B, B.T, 10, 0.8, True, 2)
        ok = (np.array_equal(all_counts.numpy(), np.diff(C_full.indptr))
              and np.array_equal(C_local.indices, C_full[lo:hi].indices)
              and np.array_equal(C_local.data, C_full[lo:hi].data))
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_broadcast_shard_gather_world2():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def _expected(names_m, names_d, top_n, thr, dtype):
    from oracle import oracle as O
    from oracle import port as P
    if names_d is None:
        (A,), _, _ = O.tfidf_sklearn(names_m, [names_m], dtype=dtype)
        B = A
    else:
        (A, B), _, _ = O.tfidf_sklearn(names_m + names_d, [names_m, names_d], dtype=dtype)
    return A, B, P.sp_matmul_topn_port(A, B.T, top_n, thr, True, 2)


def _csr_of(cols, vals, counts, n_cols):
    indptr = np.zeros(len(counts) + 1, np.int64)
    np.cumsum(counts, out=indptr[1:])
    mask = np.arange(cols.shape[1])[None, :] < counts[:, None]
    return sp.csr_matrix((vals[mask], cols[mask], indptr), shape=(len(counts), n_cols))


def _sharded_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from string_grouper_amd.synth import synth_names
        from tests._numpy_ops import NumpyOps
        ops = NumpyOps(np.float32)
        ok = {}
        # ---- self-join: every rank tokenises its block, df all-reduce, CSR all-gather, local multiply, result gather
        names = synth_names(2501, 42) + ["", "AB", "ACME HOLDINGS INC"] * 3          # odd size: uneven blocks
        lo, hi = D.row_block(rank, world, len(names))
        res, state = D.distributed_self_join(ops, names[lo:hi], 10, 0.8)
        cols, vals, counts = D.gather_topn(ops, res)
        A, _, C = _expected(names, None, 10, 0.8, np.float32)
        got = _csr_of(cols, vals, counts, len(names))
        ok["selfjoin_counts"] = np.array_equal(np.diff(got.indptr), np.diff(C.indptr))
        ok["selfjoin_indices"] = np.array_equal(got.indices, C.indices)
        ok["selfjoin_scores"] = np.array_equal(got.data, C.data)
        ok["idf"] = np.array_equal(state.idf, O_idf(names, np.float32))
        # the replicated matrix equals the single-process TF-IDF matrix bit for bit
        full = D.replicate_csr(ops, ops.transform(state, names[lo:hi]))
        ok["replicated_matrix"] = (np.array_equal(full.indptr, A.indptr) and np.array_equal(full.indices, A.indices)
                                   and np.array_equal(full.data, A.data))
        # ---- the same self-join in its form over row ranges: every rank scores the pairs (i, j <= i) of its range, the
        #      mirrored pairs are all-gathered, every rank merges those that point into its range (SG_DIST_SYM=1 forces
        #      the form at this size); hubs of duplicates make rows whose top-n is cut inside the merge
        hubs = names + [names[5]] * 40 + [names[9] + " INC"] * 25
        lo2, hi2 = D.row_block(rank, world, len(hubs))
        _, _, C = _expected(hubs, None, 10, 0.8, np.float32)
        bounds = D.selfjoin_row_ranges(len(hubs), world)
        ok["ranges_cover"] = bounds[0] == 0 and bounds[-1] == len(hubs) and bool(np.all(np.diff(bounds) > 0))
        shares = [D.share_positions(*D.selfjoin_share(len(hubs), r, world)) for r in range(world)]
        ok["interleaved_shares_cover"] = sorted(np.concatenate(shares).tolist()) == list(range(len(hubs)))

        def run_form(tag, permuted, grouped):
            ops.permuted, ops.grouped = permuted, grouped
            os.environ["SG_DIST_SYM"] = "1"
            try:
                res, _ = D.distributed_self_join(ops, hubs[lo2:hi2], 10, 0.8)
            finally:
                os.environ["SG_DIST_SYM"] = "0"
                ops.permuted = ops.grouped = False
            cols, vals, counts = D.gather_topn(ops, res)
            got = _csr_of(cols, vals, counts, len(hubs))
            ok[tag + "_counts"] = np.array_equal(np.diff(got.indptr), np.diff(C.indptr))
            ok[tag + "_indices"] = np.array_equal(got.indices, C.indices)
            ok[tag + "_scores"] = np.array_equal(got.data, C.data)
            return res

        def every_row_once(tag, res):
            ok[tag + "_block_lists_its_rows"] = res.row_ids is not None and len(res[2]) == res.row_ids.numel()
            n_mine = torch.tensor([res.row_ids.numel()])
            dist.all_reduce(n_mine)
            ok[tag + "_every_row_once"] = int(n_mine) == len(hubs)

        # (a) contiguous ranges of rows / of positions of the library's row permutation (SG_DIST_INTERLEAVE=0): a rank's block
        #     holds the rows [lo, hi) / orig_of[lo:hi], gather_topn concatenates and puts the blocks back into row order
        os.environ["SG_DIST_INTERLEAVE"] = "0"
        res = run_form("ranges", False, False)
        ok["range_block_is_mine"] = len(res[2]) == int(bounds[rank + 1] - bounds[rank])
        res = run_form("positions", True, False)
        ok["position_block_is_mine"] = len(res[2]) == int(bounds[rank + 1] - bounds[rank]) and res.orig_of is not None
        # (b) ... over one representative per group of identical rows (what the library builds by default): ranges of GROUPS,
        #     a rank's block holds the rows that are members of its groups (expanded on the rank), gather_topn puts the
        #     gathered rows where their numbers say
        for permuted in (False, True):
            every_row_once("groups_permuted" if permuted else "groups", run_form("groups_permuted" if permuted else "groups", permuted, True))
        # (c) the default: INTERLEAVED shares (rank r scores every world-th position from the top): blocks list their rows
        os.environ.pop("SG_DIST_INTERLEAVE")
        for permuted in (False, True):
            for grouped in (False, True):
                tag = "interleaved" + ("_permuted" if permuted else "") + ("_groups" if grouped else "")
                every_row_once(tag, run_form(tag, permuted, grouped))
        # ---- master x duplicates (configs[4]): both columns sharded, vocabulary from both, duplicates replicated
        master = synth_names(1800, 7)
        dups = synth_names(901, 8, perturb_of=master, perturb_frac=0.5)
        mlo, mhi = D.row_block(rank, world, len(master))
        dlo, dhi = D.row_block(rank, world, len(dups))
        res, _ = D.distributed_match(ops, master[mlo:mhi], dups[dlo:dhi], 20, 0.7)
        cols, vals, counts = D.gather_topn(ops, res)
        _, _, C = _expected(master, dups, 20, 0.7, np.float32)
        got = _csr_of(cols, vals, counts, len(dups))
        ok["match_counts"] = np.array_equal(np.diff(got.indptr), np.diff(C.indptr))
        ok["match_indices"] = np.array_equal(got.indices, C.indices)
        ok["match_scores"] = np.array_equal(got.data, C.data)
        # ---- a block that cannot share its document-frequency table on ONE rank only (that rank used to raise
        #      on its own while the other entered the all-reduce): every rank learns it from the exchange and raises alike
        for bad in ((1,), (0, 1)):
            ops.not_shareable_on = bad
            try:
                D.distributed_self_join(ops, names[lo:hi], 10, 0.8)
                ok["not_shareable_%s" % len(bad)] = False
            except D.ShardedFitNotApplicable:
                ok["not_shareable_%s" % len(bad)] = True
        ops.not_shareable_on = ()
        res, _ = D.distributed_self_join(ops, names[lo:hi], 10, 0.8)          # ... and the group is still in step
        ok["in_step_after_refusal"] = len(res[2]) == hi - lo
        # ---- gather_counts with cost-weighted (uneven) cuts
        cuts = D.weighted_row_blocks(np.arange(1, 101) ** 2, world)
        mine = torch.arange(int(cuts[rank]), int(cuts[rank + 1]), dtype=torch.int32)
        ok["weighted_gather"] = D.gather_counts(mine, 100).tolist() == list(range(100))
        ret[rank] = ok
    finally:
        dist.destroy_process_group()


def O_idf(names, dtype):
    from oracle import oracle as O
    _, _, idf = O.tfidf_sklearn(names, [names], dtype=dtype)
    return idf


@pytest.mark.timeout(600)
def test_sharded_vectoriser_gather_and_match_world2():
    """The round-2 sharded path on 2 gloo ranks: df all-reduce, CSR all-gather, self-join and master x duplicates,
    concatenation on the host -- bit-identical to the single-process oracle."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_sharded_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    for r in range(world):
        assert all(ret[r].values()), (r, dict(ret[r]))


def test_shares_of_the_self_join_form_partition_the_positions(monkeypatch):
    """distributed.selfjoin_share: interleaved by default (rank r: every world-th position counted from the top) or the
    contiguous ranges cut by cost (SG_DIST_INTERLEAVE=0) -- either way every position belongs to exactly one rank, for
    any size, also when there are fewer positions than ranks; and the descending walk of a share (the kernel's order:
    hi - 1, hi - 1 - step, ...) stays inside [lo, hi)."""
    for interleave in ("1", "0"):
        monkeypatch.setenv("SG_DIST_INTERLEAVE", interleave)
        for n in (0, 1, 2, 7, 8, 9, 63, 64, 1000, 131072, 553497):
            for world in (1, 2, 3, 5, 8):
                seen = np.zeros(n, np.int32)
                for r in range(world):
                    lo, hi, step = D.selfjoin_share(n, r, world)
                    assert 0 <= lo <= hi <= n and step >= 1, (n, world, r, lo, hi, step)
                    pos = D.share_positions(lo, hi, step)
                    walk = np.arange(hi - 1, lo - 1, -step)            # what the kernel visits
                    assert sorted(walk.tolist()) == pos.tolist(), (n, world, r)
                    assert (interleave == "1" and world > 1) == (step > 1) or n == 0 or world == 1
                    seen[pos] += 1
                assert (seen == 1).all(), (n, world, interleave)
                if interleave == "1" and world > 1 and n >= world:
                    sizes = [len(D.share_positions(*D.selfjoin_share(n, r, world))) for r in range(world)]
                    assert max(sizes) - min(sizes) <= 1            # equal shares without a cost model


def _csr_gather_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ok = True
        for dtype, cuts in ((np.float32, (0, 5, 5, 17)), (np.float64, (0, 1, 12, 17)), (np.float64, (0, 0, 0, 17))):
            # (blocks of uneven size, an EMPTY block, an odd number of non-zeros in front of an f64 block: the packed
            #  payload -- values | indices | row lengths, padded to the longest -- is cut into typed views by offset)
            rng = np.random.default_rng(3)
            M = sp.random(17, 40, density=0.3, format="csr", random_state=rng, dtype=np.float64).astype(dtype)
            M.sort_indices()
            blk = M[cuts[rank]:cuts[rank + 1]]
            ip = torch.from_numpy(blk.indptr.astype(np.int64))
            ix = torch.from_numpy(blk.indices.astype(np.int32))
            d = torch.from_numpy(blk.data.copy())
            fp, fi, fd, shape = D.all_gather_csr(ip, ix, d, 40)
            got = sp.csr_matrix((fd.numpy(), fi.numpy(), fp.numpy()), shape=shape)
            ok = ok and shape == (17, 40) and fd.numpy().dtype == dtype and np.array_equal(got.indptr, M.indptr) \
                and np.array_equal(got.indices, M.indices) and np.array_equal(got.data, M.data)
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_all_gather_csr_packs_every_block_into_one_payload_world3():
    """Round 5: the CSR blocks travel as ONE packed buffer per rank behind one header exchange (two collectives, four
    before): uneven blocks, an empty block, blocks that are all empty but one, fp32 and fp64."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_csr_gather_worker, args=(3, _free_port(), ret), nprocs=3, join=True)
    assert dict(ret) == {0: True, 1: True, 2: True}, dict(ret)
