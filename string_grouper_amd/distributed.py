"""Multi-GPU execution of the hot path: one process per GPU, ``torch.distributed`` over RCCL/xGMI.

The rows of ``C = topn(A . B^T)`` are independent and the top-n is taken per LEFT row
(string_grouper/string_grouper.py:728-729), so the left matrix is cut into contiguous row blocks,
one per rank (this is the reference's ``n_blocks[0]`` / ``vstack``, :734 and :750 -- concatenation,
no merge).  The right-hand matrix is needed by every rank: rank 0 vectorises and its CSR is
broadcast ONCE (three tensors) -- the only collective on the data path; every rank then builds the
inverted index locally (K3, a few ms) and multiplies its block (K4).  Results stay on the owning
rank; ``gather_counts`` collects the per-row match counts for reporting.

PyTorch is plumbing here: device tensors to broadcast into, and the process group.  The
orchestration below is backend-agnostic (tensors in, tensors out) so that the world_size-2 ``gloo``
tests on CPU exercise the same code path with host tensors.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np

try:
    import torch
    import torch.distributed as dist
except Exception:  # pragma: no cover - torch is part of the image
    torch = None
    dist = None


def row_block(rank: int, world: int, n_rows: int) -> Tuple[int, int]:
    """Contiguous, balanced row range of ``rank`` (sizes differ by at most one row)."""
    base, extra = divmod(n_rows, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def weighted_row_blocks(row_cost: np.ndarray, world: int) -> np.ndarray:
    """Row boundaries (world + 1 entries) that balance a per-row cost estimate, e.g. the number of
    intermediate products of each left row; used when the input order is skewed."""
    c = np.cumsum(np.asarray(row_cost, dtype=np.float64))
    total = c[-1] if len(c) else 0.0
    cuts = [0]
    for r in range(1, world):
        cuts.append(int(np.searchsorted(c, total * r / world)))
    cuts.append(len(c))
    return np.maximum.accumulate(np.asarray(cuts, dtype=np.int64))


def broadcast_csr(indptr, indices, data, shape, src: int = 0, device=None, group=None):
    """Broadcast a CSR matrix from ``src``.  On ``src`` pass torch tensors (int64 indptr, int32
    indices, float data) and ``shape``; elsewhere pass ``None`` for all four.  Returns the four on
    every rank.  One small header broadcast (sizes, dtype) + three payload broadcasts."""
    rank = dist.get_rank(group)
    header = torch.zeros(4, dtype=torch.int64, device=device)
    if rank == src:
        header[0], header[1] = int(shape[0]), int(shape[1])
        header[2] = int(indices.numel())
        header[3] = 1 if data.dtype == torch.float64 else 0
    dist.broadcast(header, src=src, group=group)
    n_rows, n_cols, nnz, is_f64 = (int(x) for x in header.tolist())
    if rank != src:
        indptr = torch.empty(n_rows + 1, dtype=torch.int64, device=device)
        indices = torch.empty(max(nnz, 1), dtype=torch.int32, device=device)[:nnz]
        data = torch.empty(max(nnz, 1), dtype=torch.float64 if is_f64 else torch.float32, device=device)[:nnz]
    dist.broadcast(indptr, src=src, group=group)
    if nnz > 0:
        dist.broadcast(indices, src=src, group=group)
        dist.broadcast(data, src=src, group=group)
    return indptr, indices, data, (n_rows, n_cols)


def gather_counts(local_counts, n_total: int, group=None):
    """All ranks' per-row match counts, concatenated in rank order (row blocks are contiguous).  The block
    sizes are exchanged first: the cuts may be the balanced ones of ``row_block`` or cost-weighted
    (``weighted_row_blocks``), and only the owning rank knows which."""
    world = dist.get_world_size(group)
    mine = torch.tensor([int(local_counts.numel())], dtype=torch.int64, device=local_counts.device)
    sizes = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(sizes, mine, group=group)
    sizes = [int(s.item()) for s in sizes]
    if sum(sizes) != n_total:
        raise ValueError(f"row blocks of the ranks hold {sum(sizes)} rows, expected {n_total}")
    longest = max(max(sizes), 1)
    padded = torch.zeros(longest, dtype=local_counts.dtype, device=local_counts.device)
    padded[: local_counts.numel()] = local_counts
    out = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(out, padded, group=group)
    return torch.cat([o[:n] for o, n in zip(out, sizes)])


class DeviceTensorView:
    """Zero-copy torch view of library-owned device memory (``__cuda_array_interface__``)."""

    def __init__(self, ptr: int, n: int, typestr: str):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


def csr_as_torch(csr) -> tuple:
    """(indptr, indices, data) torch tensors aliasing a device CSR of the HIP library."""
    import ctypes as C
    from . import _native as N
    r, c, nnz, d = csr.dims()
    p_ip, p_ix, p_d = C.c_void_p(), C.c_void_p(), C.c_void_p()
    N.check(N.lib().sg_csr_device_ptrs(csr.h, C.byref(p_ip), C.byref(p_ix), C.byref(p_d)))
    dev = torch.device("cuda", csr.ctx.device)
    indptr = torch.as_tensor(DeviceTensorView(p_ip.value, r + 1, "<i8"), device=dev)
    indices = torch.as_tensor(DeviceTensorView(p_ix.value, max(nnz, 1), "<i4"), device=dev)[:nnz]
    data = torch.as_tensor(DeviceTensorView(p_d.value, max(nnz, 1), "<f8" if d == N.SG_F64 else "<f4"), device=dev)[:nnz]
    return indptr, indices, data


def csr_from_torch(ctx, indptr, indices, data, shape):
    """Wrap broadcast tensors as a device CSR of the HIP library (no copy; tensors are kept alive)."""
    nnz = int(indices.numel())
    dtype = np.float64 if data.dtype == torch.float64 else np.float32
    return ctx.csr_from_device(shape[0], shape[1], nnz, indptr.data_ptr(), indices.data_ptr() if nnz else 0,
                               data.data_ptr() if nnz else 0, dtype, keepalive=(indptr, indices, data))


def strings_to_device_tensors(prepared, device):
    """(bytes, offsets) of a PreparedStrings as torch tensors on ``device`` (done once, outside any
    timed region; the tensors own the HBM copy that ``broadcast_strings`` sends)."""
    t_bytes = torch.from_numpy(np.ascontiguousarray(prepared.data) if prepared.data.size else
                               np.zeros(1, np.uint8)).to(device)
    t_offs = torch.from_numpy(np.ascontiguousarray(prepared.offsets)).to(device)
    return t_bytes, t_offs


def broadcast_strings(ctx, t_bytes, t_offs, src: int = 0, group=None):
    """Broadcast a string column that is resident in HBM on ``src`` (UTF-8 bytes uint8 tensor + int64
    offsets tensor; other ranks pass None, None) to every rank over RCCL and wrap it as device strings
    of the HIP library (no copy).  This is the one exchange step of the sharded path: the strings are
    ~5x smaller than the TF-IDF CSR (21 MB vs 105 MB at 663 k), and every rank can then vectorise by
    itself instead of waiting for rank ``src``."""
    from .vectorizer import PreparedStrings
    rank = dist.get_rank(group)
    dev = torch.device("cuda", ctx.device)
    header = torch.zeros(2, dtype=torch.int64, device=dev)
    if rank == src:
        header[0] = t_offs.numel() - 1
        header[1] = t_offs[-1]
    dist.broadcast(header, src=src, group=group)
    n, total = (int(x) for x in header.tolist())
    if rank != src:
        t_bytes = torch.empty(max(total, 1), dtype=torch.uint8, device=dev)
        t_offs = torch.empty(n + 1, dtype=torch.int64, device=dev)
    dist.broadcast(t_bytes, src=src, group=group)
    dist.broadcast(t_offs, src=src, group=group)
    torch.cuda.current_stream(dev).synchronize()
    p = object.__new__(PreparedStrings)
    p.data, p.offsets, p.n = None, None, n
    p.dev = ctx.strings_from_device(t_bytes.data_ptr(), t_offs.data_ptr(), n, total, keepalive=(t_bytes, t_offs))
    return p


def pruned_multiply_expected(top_n: int, threshold: float) -> bool:
    """The library's rule for taking the pruned multiply on TF-IDF input (sg_spgemm_topn).
    Its cost per left row is nearly uniform (the column-tile loop dominates), whereas the exact kernel's
    follows the row's intermediate products -- which decides how left rows are best cut across ranks."""
    import os
    if os.environ.get("SG_PRUNE", "1").startswith("0"):
        return False
    return top_n <= 64 and threshold >= float(os.environ.get("SG_PRUNE_MIN_THRESHOLD", "0.45"))


def sharded_self_join_replicated(ctx, prepared_dev, vectorizer_factory, top_n: int, threshold: float,
                                 group=None, tile_cols: int = 0, balance: Optional[bool] = None):
    """Strong-scaled self-join when every rank holds the string column in HBM (after
    ``broadcast_strings``): each rank vectorises (K1 + K2, ~4 ms at 663 k -- cheaper than receiving the
    CSR), builds the postings (K3) and multiplies ITS contiguous block of left rows (K4).  No collective
    inside.  ``balance``: cut the rows so that every rank gets the same number of intermediate products
    (sg_row_costs) instead of the same number of rows -- matters for the exact kernel when the input is
    sorted; default: only when the exact kernel will run (the pruned kernel's cost per row is uniform and
    the cost pass + its host round trip would cost more than they save).  Returns (TopN of the local
    block, (row_lo, row_hi), n_rows_total)."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if balance is None:
        balance = not pruned_multiply_expected(top_n, threshold)
    vec = vectorizer_factory()
    vec.fit_prepared([prepared_dev])
    A = vec.transform_prepared(prepared_dev)
    post = ctx.postings_build(A, tile_cols)
    n = A.dims()[0]
    if balance and world > 1:
        # every rank computes the same cuts from the same costs: no exchange needed
        cuts = weighted_row_blocks(ctx.row_costs(A, post), world)
        lo, hi = int(cuts[rank]), int(cuts[rank + 1])
    else:
        lo, hi = row_block(rank, world, n)
    block = A.row_block(lo, hi)
    res = ctx.spgemm_topn(block, post, top_n, threshold, True)
    ctx.sync()
    res._keep = (post, block, A, vec)
    return res, (lo, hi), n


def sharded_self_join(ctx, prepared_strings_or_none, vectorizer_factory, top_n: int, threshold: float,
                      group=None, tile_cols: int = 0):
    """Strong-scaled self-join, CSR-broadcast form (BASELINE.json's description of the path).

    Rank 0 vectorises (``prepared_strings_or_none`` is its PreparedStrings, other ranks pass None),
    broadcasts the TF-IDF CSR over RCCL, every rank builds the postings and multiplies its row block.
    Returns (TopN result of the local block, (row_lo, row_hi), n_rows_total)."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    dev = torch.device("cuda", ctx.device)
    if rank == 0:
        vec = vectorizer_factory()
        vec.fit_prepared([prepared_strings_or_none])
        A = vec.transform_prepared(prepared_strings_or_none)
        ctx.sync()
        t_ip, t_ix, t_d = csr_as_torch(A)
        shape = A.dims()[:2]
    else:
        A = None
        t_ip = t_ix = t_d = shape = None
    t_ip, t_ix, t_d, shape = broadcast_csr(t_ip, t_ix, t_d, shape, src=0, device=dev, group=group)
    torch.cuda.current_stream(dev).synchronize()
    B = A if rank == 0 else csr_from_torch(ctx, t_ip, t_ix, t_d, shape)
    post = ctx.postings_build(B, tile_cols)
    lo, hi = row_block(rank, world, shape[0])
    block = B.row_block(lo, hi)
    res = ctx.spgemm_topn(block, post, top_n, threshold, True)
    ctx.sync()
    res._keep = (post, block, B, t_ip, t_ix, t_d)
    return res, (lo, hi), shape[0]
