"""Multi-GPU execution of the hot path: one process per GPU, ``torch.distributed`` over RCCL/xGMI.

The rows of ``C = topn(A . B^T)`` are independent and the top-n is taken per LEFT row
(string_grouper/string_grouper.py:728-729), so the left matrix is cut into contiguous row blocks, one per
rank (the reference's ``n_blocks[0]`` / ``vstack``, :734 and :750 -- concatenation, no merge), and every rank
needs the whole right-hand matrix.  Round 2 shards the vectoriser as well (round 1 replicated it, which
bounded strong scaling at ~4x on 8 GPUs):

  1. every rank tokenises ITS block of every string column (K1) and counts document frequencies into the dense
     key table;
  2. ONE all-reduce (sum) of that table (8 MiB of int32 for 3-grams) + the document count: every rank now derives
     the same vocabulary and idf (string_grouper.py:699-707 fits on master + duplicates);
  3. every rank weights + normalises its block (K2) -> its rows of the TF-IDF matrices;
  4. ONE all-gather of the right-hand side's CSR blocks (self-join: of the matrix itself) -> every rank builds
     the inverted index (K3) and multiplies its left block (K4p / K4);
  5. results stay on the owning rank (``bench.py``), or the fixed-stride blocks are all-gathered and concatenated
     on the host for the public API (``DistributedHipEngine`` in engine.py; the north star's "per-block COO results
     concatenated on host").
No collective in the steady state of the multiply.  The self-join form of the pruned kernel (every pair scored
once) is a single-GPU optimisation: with a row block on the left the one-sided kernel runs.

The orchestration below talks to the device through an ``ops`` object (``HipOps`` here; the world_size-2 gloo
tests on CPU pass a numpy-backed double with the same methods), so the code that shards, reduces, gathers and
concatenates is the same in both.  PyTorch is plumbing: tensors that alias library memory, and the process group.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np

try:
    import torch
    import torch.distributed as dist
except Exception:  # pragma: no cover - torch is part of the image
    torch = None
    dist = None


# ---------------------------------------------------------------------------------------------- row blocks
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


# ---------------------------------------------------------------------------------------------- collectives
# Transport.  The product runs one process per GPU on RCCL (backend "nccl"), whose collectives take device tensors.  A group
# on gloo -- several ranks on ONE device (RCCL refuses two ranks of a communicator on the same GPU), which is how the device
# ops of the N > 1 path are tested and benchmarked on a one-GPU box -- has no device collectives here: the three wrappers
# below run the collective on a host copy and put the result back.  Host tensors (the CPU tests) pass through untouched.
def _host_staged(t, group) -> bool:
    return bool(t.is_cuda) and dist.get_backend(group) == "gloo"


# What the collectives of this module moved, per kind: calls and bytes this rank RECEIVED (bench.py prints them for N > 1;
# scripts/sim_scaling.py prices its model of the collectives with them).  reset_collective_tally() before a region.
COLLECTIVE_TALLY = {"all_gather": [0, 0], "all_reduce": [0, 0], "broadcast": [0, 0]}


def reset_collective_tally() -> None:
    for v in COLLECTIVE_TALLY.values():
        v[0] = v[1] = 0


def _tally(kind: str, n_bytes: int) -> None:
    COLLECTIVE_TALLY[kind][0] += 1
    COLLECTIVE_TALLY[kind][1] += int(n_bytes)


def _all_gather_into(out, t, group=None) -> None:
    _tally("all_gather", out.numel() * out.element_size() - t.numel() * t.element_size())
    if _host_staged(t, group):
        h = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(h, t.cpu(), group=group)
        out.copy_(h)
        return
    dist.all_gather_into_tensor(out, t, group=group)


def _all_reduce(t, op=None, group=None) -> None:
    op = dist.ReduceOp.SUM if op is None else op
    _tally("all_reduce", t.numel() * t.element_size())
    if _host_staged(t, group):
        h = t.cpu()
        dist.all_reduce(h, op=op, group=group)
        t.copy_(h)
        return
    dist.all_reduce(t, op=op, group=group)


def _broadcast(t, src: int, group=None) -> None:
    _tally("broadcast", t.numel() * t.element_size())
    if _host_staged(t, group):
        h = t.cpu()
        dist.broadcast(h, src=src, group=group)
        t.copy_(h)
        return
    dist.broadcast(t, src=src, group=group)


def all_headers(values: Sequence[int], device, group=None) -> List[List[int]]:
    """Every rank's small vector of integers (sizes, flags), as a list in rank order: ONE collective and ONE
    device-to-host copy -- the only synchronisation a group of ragged gathers needs."""
    world = dist.get_world_size(group)
    mine = torch.tensor([int(v) for v in values], dtype=torch.int64, device=device)
    out = torch.empty(world * mine.numel(), dtype=torch.int64, device=device)
    _all_gather_into(out, mine, group)
    return out.reshape(world, mine.numel()).tolist()


def _all_sizes(n: int, device, group=None) -> List[int]:
    return [h[0] for h in all_headers([n], device, group)]


def all_gather_ragged(t, group=None, sizes: Optional[Sequence[int]] = None) -> list:
    """All ranks' 1-d tensors (different lengths, same dtype) as a list in rank order: one padded all-gather into one
    buffer.  ``sizes``: the ranks' lengths when the caller has exchanged them already (``all_headers``: several
    gathers share one exchange); otherwise they are exchanged here."""
    if sizes is None:
        sizes = _all_sizes(t.numel(), t.device, group)
    sizes = [int(n) for n in sizes]
    longest = max(max(sizes), 1)
    if t.numel() == longest:
        padded = t.contiguous()
    else:
        padded = torch.empty(longest, dtype=t.dtype, device=t.device)
        padded[: t.numel()] = t
    out = torch.empty(len(sizes) * longest, dtype=t.dtype, device=t.device)
    _all_gather_into(out, padded, group)
    return [out[r * longest: r * longest + n] for r, n in enumerate(sizes)]


def gather_counts(local_counts, n_total: int, group=None):
    """All ranks' per-row match counts, concatenated in rank order (row blocks are contiguous).  The block
    sizes are exchanged first: the cuts may be the balanced ones of ``row_block`` or cost-weighted
    (``weighted_row_blocks``), and only the owning rank knows which."""
    parts = all_gather_ragged(local_counts, group)
    if sum(p.numel() for p in parts) != n_total:
        raise ValueError(f"row blocks of the ranks hold {sum(p.numel() for p in parts)} rows, expected {n_total}")
    return torch.cat(parts)


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
    _broadcast(header, src, group)
    n_rows, n_cols, nnz, is_f64 = (int(x) for x in header.tolist())
    if rank != src:
        indptr = torch.empty(n_rows + 1, dtype=torch.int64, device=device)
        indices = torch.empty(max(nnz, 1), dtype=torch.int32, device=device)[:nnz]
        data = torch.empty(max(nnz, 1), dtype=torch.float64 if is_f64 else torch.float32, device=device)[:nnz]
    _broadcast(indptr, src, group)
    if nnz > 0:
        _broadcast(indices, src, group)
        _broadcast(data, src, group)
    return indptr, indices, data, (n_rows, n_cols)


def all_gather_csr(indptr, indices, data, n_cols: int, group=None):
    """Concatenate the ranks' CSR row blocks (rank order = row order): returns (indptr, indices, data, shape) of
    the whole matrix on every rank.  Row pointers travel as row lengths and are rebuilt by a prefix sum.  TWO collectives
    (round 5; four before): the header exchange (rows, non-zeros of every block) and ONE all-gather of a packed buffer per
    rank -- [values | column indices | row lengths], padded to the longest rank's.  Peak memory: the gathered buffer
    (world x the longest rank's part) and the concatenated arrays made from it are alive together -- about twice the whole
    matrix -- until this function returns (round 4's three smaller gathers peaked lower; 5 M names: 2 x 0.8 GB of 288)."""
    head = all_headers([indptr.numel() - 1, indices.numel()], indptr.device, group)   # rows, non-zeros of every block
    rows, nnz = [h[0] for h in head], [h[1] for h in head]
    if indices.dtype != torch.int32:
        raise TypeError(f"all_gather_csr packs 4-byte column indices; got {indices.dtype}")
    s = data.element_size()
    sizes = [z * (s + 4) + r * 4 for r, z in zip(rows, nnz)]
    longest = (max(max(sizes), 16) + 15) // 16 * 16            # (every rank's part starts 16-byte aligned: the typed views below)
    mine = torch.empty(longest, dtype=torch.uint8, device=indptr.device)
    z, r = indices.numel(), indptr.numel() - 1
    if z > 0:       # (an empty block -- fewer rows than ranks, a block of empty strings -- has nothing to view as bytes)
        mine[: z * s] = data.contiguous().reshape(-1).view(torch.uint8)
        mine[z * s: z * (s + 4)] = indices.contiguous().reshape(-1).view(torch.uint8)
    if r > 0:
        mine[z * (s + 4): z * (s + 4) + r * 4] = (indptr[1:] - indptr[:-1]).to(torch.int32).contiguous().view(torch.uint8)   # (a row holds < 2^31 entries)
    out = torch.empty(len(rows) * longest, dtype=torch.uint8, device=indptr.device)
    _all_gather_into(out, mine, group)
    val, idx, lens = [], [], []
    for k, (r, z) in enumerate(zip(rows, nnz)):
        part = out[k * longest: (k + 1) * longest]
        val.append(part[: z * s].view(data.dtype))
        idx.append(part[z * s: z * (s + 4)].view(torch.int32))
        lens.append(part[z * (s + 4): z * (s + 4) + r * 4].view(torch.int32))
    row_len = torch.cat(lens)
    full_ptr = torch.zeros(row_len.numel() + 1, dtype=torch.int64, device=indptr.device)
    torch.cumsum(row_len, 0, out=full_ptr[1:])
    return full_ptr, torch.cat(idx), torch.cat(val), (int(row_len.numel()), int(n_cols))


# ---------------------------------------------------------------------------------------------- the sharded path
class ShardedFitNotApplicable(Exception):
    """Raised by ``sharded_tfidf`` on EVERY rank of the group alike (the decision is taken from exchanged facts, before
    any collective on the data): some rank's block is coded in a way the others cannot add up -- n-gram keys over the
    alphabet of the LOCAL strings (``ngram_size`` > 3, ``normalize_to_ascii=False`` with non-ASCII characters on some
    ranks only), or a sorted vocabulary without a dense table.  The caller vectorises the whole column on every rank
    instead (``DistributedHipEngine.tfidf``, ``sharded_self_join_replicated``)."""


def sharded_tfidf(ops, local_sets: Sequence, group=None):
    """Steps 1-3: TF-IDF of the LOCAL blocks of every string column with the vocabulary / idf of ALL ranks' strings
    (TfidfVectorizer.fit(concat(all strings)) + transform, string_grouper.py:685-707).
    ``local_sets``: this rank's block of each column, e.g. [master block] or [master block, duplicates block].
    Returns (fit state, [local CSR of each set]).  Raises ``ShardedFitNotApplicable`` -- on all ranks or on none."""
    state = ops.fit_begin(local_sets)
    world = dist.get_world_size(group)
    n_docs = sum(ops.n_strings(s) for s in local_sets)
    if world > 1:
        # Whether the document-frequency tables can be added up is a property of EVERY rank's block (a block that is
        # pure ASCII gets the shared 7-bit coding, a block with other characters its own alphabet): all ranks learn all
        # ranks' answers from one small exchange and take the same branch -- a rank that raised on its own while the
        # others entered the all-reduce would hang the job until the collective timed out.  The blocks' document counts
        # ride in the same exchange (round 5: they were an all-reduce of their own).
        shareable, entries = ops.fit_info(state)
        facts = all_headers([1 if shareable else 0, int(entries), int(n_docs)], ops.device, group)
        if not all(f[0] == 1 for f in facts) or len({f[1] for f in facts}) != 1:
            raise ShardedFitNotApplicable(
                "the n-gram keys of some rank's strings are coded over the alphabet of its LOCAL strings (ngram_size > 3, "
                "or non-ASCII characters kept by normalize_to_ascii=False), or the vocabulary is a sorted one: the document "
                f"frequencies of the ranks cannot be added up (shareable, table entries per rank: {[f[:2] for f in facts]})")
        n_docs = sum(f[2] for f in facts)
    df = ops.df_tensor(state)                                   # dense int32 table over the n-gram key space
    if world > 1:
        _all_reduce(df, dist.ReduceOp.SUM, group)
    ops.fit_end(state, int(n_docs))
    return state, [ops.transform(state, s) for s in local_sets]


def replicate_csr(ops, local_csr, group=None):
    """Step 4a: the whole matrix on every rank from the ranks' row blocks."""
    if dist.get_world_size(group) == 1:
        return local_csr
    ip, ix, d = ops.csr_tensors(local_csr)
    n_cols = ops.csr_shape(local_csr)[1]
    return ops.csr_from_tensors(*all_gather_csr(ip, ix, d, n_cols, group))


def sharded_topn(ops, left_local, right_full, top_n: int, threshold: float, tile_cols: int = 0, self_join: bool = False,
                 group=None):
    """Step 4b: inverted index of the whole right-hand side, multiply of the local left rows -- or, for a self-join
    that is large enough, the self-join form over row ranges (``sharded_selfjoin_topn``; the rank's block of the result
    is then the rows of its RANGE, which ``gather_topn`` concatenates just the same)."""
    post = ops.postings(right_full, tile_cols)
    if self_join and dist.is_initialized() and selfjoin_form_wanted(ops.csr_shape(right_full)[0], dist.get_world_size(group)):
        res = sharded_selfjoin_topn(ops, right_full, post, top_n, threshold, group)
        if res is not None:
            ops.keep_alive(res, post, left_local, right_full)
            return res
    res = ops.multiply(left_local, post, top_n, threshold)
    ops.keep_alive(res, post, left_local, right_full)
    return res


# cumulative cost of the self-join form's rows [0, x n): measured on eight ranges of the 663 k and the 5 M job, it is
# x^2.15 at both sizes (profiles/r03_sessionV_sim_scaling_5M.log, r03_sessionS_sim_scaling.log) -- a row's stream AND its
# candidates grow with its position, and the higher positions' postings no longer sit in the caches.  (Round 2 cut by
# x^2 / 2 + 0.03 x: the first of eight ranges then took 27 ms where the last took 38.)
SELFJOIN_COST_EXPONENT = 2.15


def selfjoin_row_ranges(n_rows: int, world: int) -> np.ndarray:
    """Left-row ranges of the self-join form across ranks: rank r scores the rows [b[r], b[r + 1]) against the columns
    j <= i, so a row's cost grows with its index: cumulative cost ~ x^2.15, cut into equal shares."""
    share = np.arange(world + 1, dtype=np.float64) / world
    x = share ** (1.0 / SELFJOIN_COST_EXPONENT)
    b = np.minimum(np.round(x * n_rows).astype(np.int64), n_rows)
    b[0], b[-1] = 0, n_rows
    return np.maximum.accumulate(b)


def selfjoin_share(n_rows: int, rank: int, world: int):
    """(lo, hi, step): the positions rank ``rank`` scores in the self-join form -- hi - 1, hi - 1 - step, ... >= lo
    (include/sg_hip.h: sg_selfjoin_range).  Interleaved by default: rank r takes (0, n - r, world), i.e. every world-th
    position counted from the top.  A position's cost grows with the position; interleaved, every rank holds rows of
    every cost -- equal shares without a cost model -- and its launch ends with its cheapest rows, as the whole pass on
    one GPU does; a contiguous range of high positions ends with rows as expensive as its first, which cost + 0.8 ms per
    range at 663 k (scripts/sim_scaling.py).  ``SG_DIST_INTERLEAVE=0``: the contiguous ranges of ``selfjoin_row_ranges``."""
    import os
    if world <= 1:
        return 0, n_rows, 1
    if os.environ.get("SG_DIST_INTERLEAVE", "1") == "0":
        b = selfjoin_row_ranges(n_rows, world)
        return int(b[rank]), int(b[rank + 1]), 1
    return 0, max(n_rows - rank, 0), world


def share_positions(lo: int, hi: int, step: int) -> np.ndarray:
    """The positions of a share, in the order the blocks list their rows (ascending)."""
    if step <= 1:
        return np.arange(lo, hi, dtype=np.int64)
    return np.arange(hi - 1, lo - 1, -step, dtype=np.int64)[::-1].copy()


def selfjoin_form_wanted(n_rows: int, world: int) -> bool:
    """From this size on the self-join form (every pair scored once, mirrored pairs exchanged by one all-gather) beats
    the one-sided multiply of row blocks; ``SG_DIST_SYM=0|1`` forces it off / on."""
    import os
    flag = os.environ.get("SG_DIST_SYM", "")
    if flag in ("0", "1"):
        return flag == "1"                     # (forced on even on one rank: the plumbing test)
    return world > 1 and n_rows >= int(os.environ.get("SG_DIST_SYM_MIN_ROWS", "131072"))


def sharded_selfjoin_topn(ops, A_full, post, top_n: int, threshold: float, group=None):
    """The self-join form across ranks (include/sg_hip.h: sg_selfjoin_range / sg_selfjoin_merge).  Every rank scores
    the pairs (i, j <= i) of ITS row range on the replicated matrix, keeps its rows' own matches, and publishes the
    mirrored pairs; ONE all-gather later every rank merges the pairs that point into its range.  Returns the rank's
    block of the result (rows of its range, in rank order = row order), or None when the form does not apply to the
    input on some rank (then nothing has been changed and the caller multiplies row blocks)."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    n = ops.selfjoin_rows(A_full, post)        # (rows of the INDEX: groups of identical rows, when the library grouped them)
    lo, hi, step = selfjoin_share(n, rank, world)
    part = ops.selfjoin_range(A_full, post, top_n, threshold, lo, hi, step)
    pairs = ops.selfjoin_pairs(part) if part is not None else None
    # one exchange tells every rank the lengths of all pair lists AND whether every range was applicable (-1: not) -- and
    # that all ranks built the same index: ranges and merge are in the index's position space, so an index over the row
    # permutation on one rank and in row order on another (or over groups on one only) would give rows that look fine
    permuted = int(bool(ops.index_is_permuted(post))) if hasattr(ops, "index_is_permuted") else 0
    heads = all_headers([pairs.numel() if pairs is not None else -1, n, permuted], ops.device, group)
    if len({(h[1], h[2]) for h in heads}) != 1:
        raise RuntimeError(f"ranks built different indexes (rows of the index, permuted): {[(h[1], h[2]) for h in heads]}")
    sizes = [h[0] for h in heads]
    if min(sizes) < 0:
        if part is not None:
            ops.selfjoin_discard(part)
        return None
    pairs_all = torch.cat(all_gather_ragged(pairs, group, sizes))
    return ops.selfjoin_merge(part, pairs_all, lo, hi, step)


def gather_topn(ops, res, group=None, on_device: bool = False):
    """Step 5 for the public API: every rank's fixed-stride block, all-gathered and concatenated in row order (the
    reference's vstack, string_grouper.py:750).  Returns (cols [n, stride], vals [n, stride], counts [n]), identical on
    every rank: numpy arrays on the HOST, or -- ``on_device`` -- the torch tensors in HBM as they come out of the
    all-gather (the fused tail of fit(), K6-K8, runs on the device: no reason to cross PCIe twice)."""
    cols, vals, counts = ops.topn_tensors(res)
    stride = cols.shape[1] if cols.dim() == 2 else 1
    row_ids = getattr(res, "row_ids", None)
    if row_ids is not None and dist.get_world_size(group) > 1:
        # the rank's block holds the rows row_ids (members of its groups of identical rows): their numbers travel along
        sizes = [h[0] for h in all_headers([row_ids.numel()], counts.device, group)]
        row_ids = torch.cat(all_gather_ragged(row_ids, group, sizes))
    if dist.get_world_size(group) > 1:
        head = all_headers([stride, counts.numel()], counts.device, group)
        strides, rows = [h[0] for h in head], [h[1] for h in head]
        if len(set(strides)) != 1:
            raise RuntimeError(f"ranks disagree on the result stride: {strides}")
        cells = [r * stride for r in rows]
        cols = torch.cat(all_gather_ragged(cols.reshape(-1), group, cells)).reshape(-1, stride)
        vals = torch.cat(all_gather_ragged(vals.reshape(-1), group, cells)).reshape(-1, stride)
        counts = torch.cat(all_gather_ragged(counts, group, rows))
    orig_of = getattr(res, "orig_of", None)
    if orig_of is not None:
        # the ranks' ranges were ranges of positions of the library's row permutation: block p of the concatenation is
        # row orig_of[p]
        if counts.numel() != orig_of.numel():
            raise RuntimeError(f"the ranks' ranges hold {counts.numel()} rows, the permutation {orig_of.numel()}")
        cols = torch.empty_like(cols).index_copy_(0, orig_of, cols)
        vals = torch.empty_like(vals).index_copy_(0, orig_of, vals)
        counts = torch.empty_like(counts).index_copy_(0, orig_of, counts)
    if row_ids is not None:
        if counts.numel() != row_ids.numel() or (row_ids.numel() and int(row_ids.max()) >= row_ids.numel()):
            raise RuntimeError(f"the ranks' blocks hold {counts.numel()} rows for {row_ids.numel()} row numbers")
        row_ids = row_ids.to(torch.int64)
        cols = torch.empty_like(cols).index_copy_(0, row_ids, cols)
        vals = torch.empty_like(vals).index_copy_(0, row_ids, vals)
        counts = torch.empty_like(counts).index_copy_(0, row_ids, counts)
    if on_device:
        return cols.contiguous(), vals.contiguous(), counts.contiguous()
    return cols.cpu().numpy(), vals.cpu().numpy(), counts.cpu().numpy()


def distributed_self_join(ops, local_block, top_n: int, threshold: float, group=None, tile_cols: int = 0):
    """Self-join of a string column of which this rank holds ``local_block`` (rows in rank order).
    Returns (TopN of the local rows -- columns index the WHOLE column --, fit state)."""
    state, (A_local,) = sharded_tfidf(ops, [local_block], group)
    A_full = replicate_csr(ops, A_local, group)
    return sharded_topn(ops, A_local, A_full, top_n, threshold, tile_cols, self_join=True, group=group), state


def distributed_match(ops, master_block, duplicates_block, top_n: int, threshold: float, group=None,
                      tile_cols: int = 0):
    """master x duplicates (BASELINE.json configs[4]; string_grouper.py:286-290, :728-729: rows = master): this rank
    holds a block of BOTH columns; the vocabulary is fitted on all of both; the master block stays where it is,
    the duplicates' rows are replicated; every rank multiplies its master rows with all duplicates."""
    state, (A_local, B_local) = sharded_tfidf(ops, [master_block, duplicates_block], group)
    B_full = replicate_csr(ops, B_local, group)
    return sharded_topn(ops, A_local, B_full, top_n, threshold, tile_cols), state


# ---------------------------------------------------------------------------------------------- HIP adapter
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
    """Wrap tensors as a device CSR of the HIP library (no copy; tensors are kept alive)."""
    nnz = int(indices.numel())
    dtype = np.float64 if data.dtype == torch.float64 else np.float32
    indptr, indices, data = indptr.contiguous(), indices.contiguous(), data.contiguous()
    return ctx.csr_from_device(shape[0], shape[1], nnz, indptr.data_ptr(), indices.data_ptr() if nnz else 0,
                               data.data_ptr() if nnz else 0, dtype, keepalive=(indptr, indices, data))


class TopNRows:
    """The rows [lo, hi) of a device result (a rank's block of the self-join form over row ranges).  ``orig_of`` given:
    the range is one of POSITIONS of the library's row permutation (int64 tensor, position -> row): the block's rows are
    orig_of[lo:hi], in that order; the blocks of all ranks concatenated are the result in position order."""

    def __init__(self, res, lo: int, hi: int, orig_of=None, row_ids=None, sel=None):
        self.res, self.lo, self.hi, self.orig_of = res, lo, hi, orig_of
        # ``row_ids`` given (int tensor): block row k is the caller's row row_ids[k] -- the rank's share was not a contiguous
        # range of rows (interleaved shares; an index over GROUPS of identical rows, whose members are the block's rows);
        # ``sel`` given (int64 tensor): the block is the rows sel of ``res`` (else the rows [lo, hi))
        self.row_ids, self.sel = row_ids, sel
        self._keep = None

    def free(self):
        self.res.free()

    def dims(self):
        r, s, d, c = self.res.dims()
        return (int(self.sel.numel()) if self.sel is not None else self.hi - self.lo), s, d, c

    def to_host(self):
        cols, vals, cnt = self.res.to_host()
        if self.sel is not None:
            ids = self.sel.cpu().numpy()
            return cols[ids], vals[ids], cnt[ids]
        if self.orig_of is not None:
            ids = self.orig_of[self.lo:self.hi].cpu().numpy()
            return cols[ids], vals[ids], cnt[ids]
        return cols[self.lo:self.hi], vals[self.lo:self.hi], cnt[self.lo:self.hi]

    def to_scipy(self):
        import scipy.sparse as sp
        cols, vals, cnt = self.to_host()
        indptr = np.zeros(len(cnt) + 1, np.int64)
        np.cumsum(cnt, out=indptr[1:])
        mask = np.arange(cols.shape[1], dtype=np.int32)[None, :] < cnt[:, None]
        return sp.csr_matrix((vals[mask], cols[mask], indptr), shape=(len(cnt), self.res.dims()[3]))


class HipOps:
    """The device side of the sharded path on the MI355X library."""

    def __init__(self, ctx, vectorizer_factory):
        self.ctx = ctx
        self.make_vec = vectorizer_factory
        self.device = torch.device("cuda", ctx.device)

    def _sync(self):
        # the library launches on its own stream (or torch's current one): hand-offs between the two are ordered
        # by a full synchronisation -- a handful per fit(), each a few microseconds when nothing is pending
        self.ctx.sync()
        torch.cuda.current_stream(self.device).synchronize()

    def n_strings(self, prepared) -> int:
        return prepared.n

    def fit_begin(self, local_sets):
        vec = self.make_vec()
        vec.fit_begin_prepared(list(local_sets))
        return vec

    def df_tensor(self, vec):
        ptr, n, _ = vec.df_table()
        self._sync()
        return torch.as_tensor(DeviceTensorView(ptr, n, "<i4"), device=self.device)

    def df_shareable(self, vec) -> bool:
        return vec.df_table()[2]

    def fit_info(self, vec):
        """(the table can be added to other ranks' tables, its entries) -- without touching the table (a sorted vocabulary
        has none)."""
        _, n, ok = vec.df_table()
        return bool(ok), int(n)

    def fit_end(self, vec, n_docs_total: int):
        self._sync()
        vec.fit_end(n_docs_total)

    def transform(self, vec, prepared):
        return vec.transform_prepared(prepared)

    def csr_tensors(self, csr):
        self._sync()
        return csr_as_torch(csr)

    def csr_shape(self, csr):
        return csr.dims()[:2]

    def csr_from_tensors(self, indptr, indices, data, shape):
        self._sync()
        return csr_from_torch(self.ctx, indptr, indices, data, shape)

    def postings(self, csr, tile_cols: int = 0, permute: bool = True):
        return self.ctx.postings_build(csr, tile_cols, permute)

    def multiply(self, left, post, top_n, threshold):
        return self.ctx.spgemm_topn(left, post, top_n, threshold, True)

    def keep_alive(self, res, *objs):
        res._keep = objs

    # ---- the self-join form over row ranges (sharded_selfjoin_topn)
    def selfjoin_range(self, A_full, post, top_n, threshold, lo, hi, step=1):
        got = self.ctx.selfjoin_range(A_full, post, top_n, threshold, lo, hi, step)
        if got is None:
            return None
        res, ptr, n_pairs, words = got
        return {"res": res, "ptr": ptr, "n": n_pairs, "words": words, "post": post}

    def selfjoin_pairs(self, part):
        self._sync()
        n = part["n"] * part["words"]
        if n == 0:
            return torch.zeros(0, dtype=torch.int32, device=self.device)
        return torch.as_tensor(DeviceTensorView(part["ptr"], n, "<i4"), device=self.device)[:n]

    def selfjoin_discard(self, part):
        self.ctx.device_free(part["ptr"])
        part["res"].free()

    def selfjoin_merge(self, part, pairs_all, lo, hi, step=1):
        self._sync()                                  # the gathered list is torch's: ordered before the library reads it
        words = part["words"]
        pairs_all = pairs_all.contiguous()
        post = part.get("post")
        self.ctx.selfjoin_merge(part["res"], post, pairs_all.data_ptr(), pairs_all.numel() // words, words, lo, hi, step)
        self.ctx.sync()                               # ... and the library is done with it before torch frees it
        self.ctx.device_free(part["ptr"])
        # the index is built over a permutation of the rows: the range [lo, hi) is one of POSITIONS, its rows are orig_of[lo:hi]
        orig_of = None
        if post is not None:
            p_orig, _ = self.ctx.postings_permutation(post)
            if p_orig:
                n = part["res"].dims()[0]
                orig_of = torch.as_tensor(DeviceTensorView(p_orig, n, "<i4"), device=self.device)[:n].to(torch.int64)
        n_index, n_caller, p_gid = self.ctx.postings_rows(post) if post is not None else (0, 0, 0)
        if not p_gid and step > 1:
            # an interleaved share: the block's rows are the rows at the share's positions
            pos = torch.from_numpy(share_positions(lo, hi, step)).to(self.device)
            rows = orig_of.index_select(0, pos) if orig_of is not None else pos
            return TopNRows(part["res"], 0, int(rows.numel()), None, row_ids=rows, sel=rows)
        if not p_gid:
            return TopNRows(part["res"], lo, hi, orig_of)
        # an index over groups of identical rows: the range was one of groups; the rows of this rank are the members of
        # its groups, expanded here from tables every rank holds (no exchange)
        r, p_rows, n_mine = self.ctx.topn_expand_range(post, part["res"], lo, hi, step)
        self.ctx.sync()
        rows = torch.as_tensor(DeviceTensorView(p_rows, max(n_mine, 1), "<i4"), device=self.device)[:n_mine].clone()
        torch.cuda.current_stream(self.device).synchronize()   # (copied before the library's list is released)
        self.ctx.device_free(p_rows)
        part["res"].free()
        return TopNRows(r, 0, rows.numel(), None, row_ids=rows)

    def selfjoin_rows(self, A_full, post):
        return self.ctx.postings_rows(post)[0]

    def index_is_permuted(self, post):
        return self.ctx.postings_permutation(post)[0] != 0

    def topn_from_tensors(self, cols, vals, counts, n_cols: int):
        """The gathered result as a library object, without leaving HBM."""
        self._sync()                                  # the tensors are torch's: written before the library copies them
        r = self.ctx.topn_from_device(cols.shape[0], cols.shape[1] if cols.dim() == 2 else 1, n_cols,
                                      np.float64 if vals.dtype == torch.float64 else np.float32,
                                      cols.data_ptr(), vals.data_ptr(), counts.data_ptr())
        self.ctx.sync()                               # ... and copied before torch may free them
        return r

    def topn_tensors(self, res):
        import ctypes as C
        from . import _native as N
        if isinstance(res, TopNRows):
            cols, vals, counts = self.topn_tensors(res.res)
            if res.sel is not None:
                return cols.index_select(0, res.sel), vals.index_select(0, res.sel), counts.index_select(0, res.sel)
            if res.orig_of is not None:               # rows of the range in POSITION order
                ids = res.orig_of[res.lo:res.hi]
                return cols.index_select(0, ids), vals.index_select(0, ids), counts.index_select(0, ids)
            return cols[res.lo:res.hi], vals[res.lo:res.hi], counts[res.lo:res.hi]
        r, s, d, _ = res.dims()
        pc, pv, pn = C.c_void_p(), C.c_void_p(), C.c_void_p()
        N.check(N.lib().sg_topn_device_ptrs(res.h, C.byref(pc), C.byref(pv), C.byref(pn)))
        self._sync()
        cols = torch.as_tensor(DeviceTensorView(pc.value, max(r * s, 1), "<i4"), device=self.device)[:r * s].reshape(r, s)
        vals = torch.as_tensor(DeviceTensorView(pv.value, max(r * s, 1), "<f8" if d == N.SG_F64 else "<f4"),
                               device=self.device)[:r * s].reshape(r, s)
        counts = torch.as_tensor(DeviceTensorView(pn.value, max(r, 1), "<i4"), device=self.device)[:r]
        return cols, vals, counts


# ---------------------------------------------------------------------------------------------- string exchange
def strings_to_device_tensors(prepared, device):
    """(bytes, offsets) of a PreparedStrings as torch tensors on ``device`` (done once, outside any
    timed region; the tensors own the HBM copy that ``broadcast_strings`` sends)."""
    t_bytes = torch.from_numpy(np.ascontiguousarray(prepared.data) if prepared.data.size else
                               np.zeros(1, np.uint8)).to(device)
    t_offs = torch.from_numpy(np.ascontiguousarray(prepared.offsets)).to(device)
    return t_bytes, t_offs


def _wrap_device_strings(ctx, t_bytes, t_offs, n: int, total: int):
    from .strprep import StringColumn
    p = StringColumn("bytes", None, np.zeros(n + 1, np.int64))       # the bytes live in HBM only
    p.offsets = None
    p.dev = ctx.strings_from_device(t_bytes.data_ptr(), t_offs.data_ptr(), n, total, keepalive=(t_bytes, t_offs))
    return p


def local_string_block(ctx, t_bytes, t_offs, rank: int, world: int):
    """This rank's contiguous block of a string column that is resident in HBM (bytes + int64 offsets tensors of the
    WHOLE column): a view, offsets rebased on the device.  Returns (PreparedStrings of the block, (lo, hi))."""
    n = int(t_offs.numel()) - 1
    lo, hi = row_block(rank, world, n)
    b0, b1 = int(t_offs[lo].item()), int(t_offs[hi].item())
    offs = (t_offs[lo:hi + 1] - b0).contiguous()
    bytes_ = t_bytes[b0:max(b1, b0 + 1)].contiguous() if b1 > b0 else torch.zeros(1, dtype=torch.uint8, device=t_bytes.device)
    torch.cuda.current_stream(t_bytes.device).synchronize()
    return _wrap_device_strings(ctx, bytes_, offs, hi - lo, b1 - b0), (lo, hi)


def broadcast_strings(ctx, t_bytes, t_offs, src: int = 0, group=None):
    """Broadcast a string column that is resident in HBM on ``src`` (UTF-8 bytes uint8 tensor + int64
    offsets tensor; other ranks pass None, None) to every rank over RCCL.  Returns (PreparedStrings of the whole
    column, bytes tensor, offsets tensor).  21 MB at 663 k names -- a fifth of the TF-IDF CSR."""
    rank = dist.get_rank(group)
    dev = torch.device("cuda", ctx.device)
    header = torch.zeros(2, dtype=torch.int64, device=dev)
    if rank == src:
        header[0] = t_offs.numel() - 1
        header[1] = t_offs[-1]
    _broadcast(header, src, group)
    n, total = (int(x) for x in header.tolist())
    if rank != src:
        t_bytes = torch.empty(max(total, 1), dtype=torch.uint8, device=dev)
        t_offs = torch.empty(n + 1, dtype=torch.int64, device=dev)
    _broadcast(t_bytes, src, group)
    _broadcast(t_offs, src, group)
    torch.cuda.current_stream(dev).synchronize()
    return _wrap_device_strings(ctx, t_bytes, t_offs, n, total), t_bytes, t_offs


# ---------------------------------------------------------------------------------------------- round-1 forms
def pruned_multiply_expected(top_n: int, threshold: float, ctx=None) -> bool:
    """The library's rule for taking the pruned multiply on TF-IDF input (sg_spgemm_topn).
    Its cost per left row is nearly uniform (the column-tile loop dominates), whereas the exact kernel's
    follows the row's intermediate products -- which decides how left rows are best cut across ranks.
    ``ctx``: the context whose switches apply (its frozen copy of SG_PRUNE / SG_PRUNE_MIN_THRESHOLD)."""
    import os
    opts = ctx.options() if ctx is not None else os.environ
    if opts.get("SG_PRUNE", "1").startswith("0"):
        return False
    # (2 x SG_TOPN_LANES: pruned_applicable; 0.40: the tile-by-tile form's envelope, prune_min_threshold in sg_spgemm_topn.hip)
    alt = not opts.get("SG_ALT_FORM", "1").startswith("0")
    return top_n <= 128 and threshold >= float(opts.get("SG_PRUNE_MIN_THRESHOLD", "0.40" if alt else "0.45"))


def sharded_self_join_replicated(ctx, prepared_dev, vectorizer_factory, top_n: int, threshold: float,
                                 group=None, tile_cols: int = 0, balance: Optional[bool] = None):
    """Round 1's form: every rank holds the whole string column, vectorises ALL of it (replicated K1 + K2), builds
    the postings and multiplies its block of left rows.  No collective inside; kept for ngram_size > 3 (whose
    character coding cannot be shared) and for cost-balanced cuts with the exact kernel.  ``balance``: cut the rows
    so that every rank gets the same number of intermediate products (sg_row_costs) instead of the same number of
    rows.  Returns (TopN of the local block, (row_lo, row_hi), n_rows_total)."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if balance is None:
        balance = not pruned_multiply_expected(top_n, threshold, ctx)
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
    """Strong-scaled self-join, CSR-broadcast form (BASELINE.json's literal description of the path): rank 0
    vectorises (``prepared_strings_or_none`` is its PreparedStrings, other ranks pass None), broadcasts the TF-IDF
    CSR over RCCL, every rank builds the postings and multiplies its row block.
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
