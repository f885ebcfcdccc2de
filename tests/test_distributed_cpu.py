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
