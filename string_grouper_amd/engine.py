"""Compute engine behind the StringGrouper front end: the MI355X library, nothing else.

The front end (string_grouper_amd/string_grouper.py) talks to an engine object with four
operations -- vectorise, wrap a host matrix, top-n multiply, blocked top-n multiply -- so that the
host logic can be unit-tested on a machine without a GPU by injecting a test double
(tests/_oracle_engine.py).  The product ships exactly one engine, ``HipEngine``; there is no CPU
engine in this package and ``get_engine()`` raises if the HIP library cannot be used.
"""
from __future__ import annotations

import os
import time
from typing import List, Optional, Sequence, Tuple

import numpy as np
import scipy.sparse as sp

from . import _hostops
from . import _native as N
from .vectorizer import HipTfidfVectorizer


class DeviceMatrix:
    """A CSR matrix resident in HBM (rows = strings, columns = n-grams)."""

    def __init__(self, csr: "N.Csr"):
        self.csr = csr
        r, c, nnz, d = csr.dims()
        self.shape = (r, c)
        self.nnz = nnz
        self.dtype = N.code_np_dtype(d)
        self._host: Optional[sp.csr_matrix] = None

    def to_scipy(self) -> sp.csr_matrix:
        if self._host is None:
            self._host = self.csr.to_scipy()
        return self._host


def chunk_ranges(length: int, n_chunks: int) -> List[Tuple[int, int]]:
    """Contiguous ranges of ceil(length / n_chunks) rows (the reference's define_chunks,
    string_grouper.py:714-722)."""
    size = int(np.ceil(length / n_chunks))
    return [(lo, min(lo + size, length)) for lo in range(0, length, size)] if length > 0 else []


class DeviceMatchList:
    """The match list of one fit(), resident in HBM, for the reductions the reference runs over it:
    best master per duplicate (K7) and group representatives (K8)."""

    def __init__(self, ml: "N.MatchList", n_cols: int):
        self.ml = ml
        self.n_cols = n_cols

    def best_master(self) -> np.ndarray:
        return self.ml.best_master(self.n_cols)

    def group_reps(self, centroid: bool) -> np.ndarray:
        return self.ml.group_reps(centroid)

    def free(self):
        if self.ml is not None:
            self.ml.free()
            self.ml = None


class HipEngine:
    name = "hip"

    def __init__(self, ctx: Optional[N.Context] = None):
        self._ctx = ctx
        # wall-clock split of the most recent fit() through this engine (seconds): host string preparation +
        # upload, device vectorise, device multiply + match list, download.  Read by bench.py (end_to_end).
        self.timings = {}

    def _tick(self, name: str, t0: float) -> float:
        now = time.perf_counter()
        self.timings[name] = self.timings.get(name, 0.0) + now - t0
        return now

    @property
    def ctx(self) -> N.Context:
        if self._ctx is None:
            self._ctx = N.default_context()
        return self._ctx

    # ------------------------------------------------------------------ seam b1
    def tfidf(self, master, duplicates, ngram_size, regex, ignore_case, normalize_to_ascii, dtype):
        """fit on master (+ duplicates), transform both (string_grouper.py:685-707).  One
        tokenisation pass per series: transform reuses the tokens of fit."""
        vec = HipTfidfVectorizer(ngram_size=ngram_size, regex=regex, ignore_case=ignore_case,
                                 normalize_to_ascii=normalize_to_ascii, dtype=dtype, ctx=self.ctx)
        self.timings = {}
        t = time.perf_counter()
        pm = vec.prepare(master)
        sets = [pm]
        if duplicates is not None:
            pd_ = vec.prepare(duplicates)
            sets.append(pd_)
        t = self._tick("prepare_and_upload_s", t)
        vec.fit_prepared(sets)
        A = DeviceMatrix(vec.transform_prepared(pm))
        B = A if duplicates is None else DeviceMatrix(vec.transform_prepared(sets[1]))
        self._tick("vectorise_s", t)             # fit_prepared reads the vocabulary back: a synchronisation point
        return A, B, vec

    def wrap(self, m) -> DeviceMatrix:
        if isinstance(m, DeviceMatrix):
            return m
        m = sp.csr_matrix(m)
        if m.dtype not in (np.float32, np.float64):
            m = m.astype(np.float64)
        d = DeviceMatrix(self.ctx.csr_from_scipy(m))
        return d

    # ------------------------------------------------------------------ seam b2
    def _topn_device(self, A: DeviceMatrix, B: DeviceMatrix, top_n: int, threshold: float) -> "N.TopN":
        """Top-n multiply with the result left on the device.  One inverted index normally; when the
        right-hand side is too large for one (SG_ERR_OVERFLOW -> OverflowError, what the reference's
        fit() reacts to by splitting, string_grouper.py:397-413) it is cut into the fewest row blocks that
        fit and the partial results are merged on the device (K5 = zip_sp_matmul_topn)."""
        ctx = self.ctx
        try:
            post = ctx.postings_build(B.csr)
        except OverflowError:
            post = None
        if post is not None:
            res = ctx.spgemm_topn(A.csr, post, top_n, threshold, True)
            post.free()
            return res
        n_blocks = 2
        while True:
            ranges = chunk_ranges(B.shape[0], n_blocks)
            views, posts = [], []
            try:
                for lo, hi in ranges:
                    v = B.csr.row_block(lo, hi)
                    views.append(v)
                    posts.append(ctx.postings_build(v))
                break
            except OverflowError:
                for h in posts + views:
                    h.free()
                n_blocks *= 2
                if n_blocks > max(B.shape[0], 2):
                    raise
        parts = [ctx.spgemm_topn(A.csr, p, top_n, threshold, True) for p in posts]
        res = ctx.topn_zip(parts, np.array([lo for lo, _ in ranges], dtype=np.int64), top_n)
        for h in parts + posts + views:
            h.free()
        return res

    def topn_multiply(self, A: DeviceMatrix, B: DeviceMatrix, top_n: int, threshold: float) -> sp.csr_matrix:
        """sp_matmul_topn(A, B.T, top_n, threshold, sort=True) (string_grouper.py:725-732)."""
        res = self._topn_device(A, B, top_n, threshold)
        C = res.to_scipy()
        C = sp.csr_matrix((C.data, C.indices, C.indptr), shape=(A.shape[0], B.shape[0]))
        res.free()
        return C

    def rowwise_dot(self, A: DeviceMatrix, B: DeviceMatrix) -> np.ndarray:
        """Row-wise similarity of master and duplicates (string_grouper.py:433-440) on the device (K9)."""
        return self.ctx.rowwise_dot(A.csr, B.csr)

    # ------------------------------------------------------------------ fused tail of fit() (K6)
    def match_list(self, A: DeviceMatrix, B: DeviceMatrix, top_n: int, threshold: float, self_join_fix: bool,
                   keep_on_device: bool = False):
        """Multiply and build the match list without leaving the device: (master_side, dupe_side,
        similarity, true_max_n_matches).  ``self_join_fix``: set the diagonal to 1 and symmetrise
        (string_grouper.py:419-427); the rows then come back sorted by column.  ``keep_on_device``: a fifth
        element, the device-resident list (``DeviceMatchList``), for the reductions over it (K7, K8)."""
        t = time.perf_counter()
        res = self._topn_device(A, B, top_n, threshold)
        return self._match_list_from_topn(res, A.dtype, B.shape[0], self_join_fix, keep_on_device, t)

    def _match_list_from_topn(self, res: "N.TopN", dtype, n_cols: int, self_join_fix: bool, keep_on_device: bool,
                              t: float):
        cnt = res.counts()
        true_max = int(cnt.max()) if len(cnt) else 0
        t = self._tick("multiply_s", t)           # counts() waits for the multiply
        # the reference up-casts a float32 result to float64 through scipy (vstack(dtype=float64), :750),
        # which re-sorts every row by column; a float64 result keeps the multiply's score-descending order
        ml = self.ctx.matchlist_build(res, self_join_fix, self_join_fix,
                                      sort_by_column=(not self_join_fix) and dtype == np.float32)
        if os.environ.get("SG_E2E_SPLIT"):            # (diagnostic: the list's kernels and its download apart)
            self.ctx.sync()
            t_ml = time.perf_counter()
            self.timings["match_list_kernels_s"] = t_ml - t
        row_ptr, cols, vals = ml.to_host()
        t = self._tick("match_list_and_download_s", t)
        res.free()
        # (round 6: the three expansions on host threads -- _hostops; numpy's own below 262 144 entries or without the helpers)
        rows = _hostops.expand_rows(row_ptr)
        cols = _hostops.widen(cols, np.int64)
        if keep_on_device:
            return rows, cols, vals, true_max, DeviceMatchList(ml, n_cols)
        ml.free()
        return rows, cols, vals, true_max

    def topn_multiply_blocked(self, A: DeviceMatrix, B: DeviceMatrix, n_blocks: Tuple[int, int], top_n: int,
                              threshold: float) -> sp.csr_matrix:
        """Block-pair products, zipped over right blocks, stacked over left blocks
        (string_grouper.py:733-752)."""
        ctx = self.ctx
        a_ranges = chunk_ranges(A.shape[0], n_blocks[0])
        b_ranges = chunk_ranges(B.shape[0], n_blocks[1])
        b_views = [B.csr.row_block(lo, hi) for lo, hi in b_ranges]
        b_posts = [ctx.postings_build(v) for v in b_views]
        offsets = np.array([lo for lo, _ in b_ranges], dtype=np.int64)
        stacked = []
        for lo, hi in a_ranges:
            a_view = A.csr.row_block(lo, hi)
            parts = [ctx.spgemm_topn(a_view, p, top_n, threshold, True) for p in b_posts]
            if len(parts) == 1:
                stacked.append(parts[0].to_scipy())
            else:
                z = ctx.topn_zip(parts, offsets, top_n)
                C = z.to_scipy()
                stacked.append(sp.csr_matrix((C.data, C.indices, C.indptr), shape=(hi - lo, B.shape[0])))
                z.free()
            for p in parts:
                p.free()
            a_view.free()
        for p in b_posts:
            p.free()
        for v in b_views:
            v.free()
        if not stacked:
            return sp.csr_matrix((A.shape[0], B.shape[0]), dtype=np.float64)
        return sp.vstack(stacked, dtype=np.float64).tocsr()


class ShardedMatrix(DeviceMatrix):
    """This rank's contiguous row block of a TF-IDF matrix whose other rows live on the other ranks.  ``shape`` is
    the shape of the WHOLE matrix (what the front end reasons about); ``csr`` holds rows [lo, hi)."""

    def __init__(self, csr: "N.Csr", n_total: int, row_range: Tuple[int, int], ops, group):
        super().__init__(csr)
        self.local_shape = self.shape
        self.shape = (n_total, self.shape[1])
        self.row_range = row_range
        self._ops, self._group = ops, group
        self._full: Optional["N.Csr"] = None

    def full(self) -> "N.Csr":
        """The whole matrix on this rank (one all-gather of the ranks' blocks, cached)."""
        if self._full is None:
            from . import distributed as D
            self._full = D.replicate_csr(self._ops, self.csr, self._group)
        return self._full

    def to_scipy(self) -> sp.csr_matrix:
        if self._host is None:
            self._host = self.full().to_scipy()
        return self._host


class DistributedHipEngine(HipEngine):
    """The engine behind ``fit()`` when the process is one rank of a ``torch.distributed`` group (one process per
    GPU, RCCL): every rank runs the SAME script on the SAME input Series and gets the SAME results.  The ranks split
    the rows of every string column (vectorise), all-reduce the document frequencies, all-gather the right-hand
    TF-IDF rows, multiply their block of left rows and all-gather the fixed-stride results, which are concatenated
    on the host (string_grouper.py:750 vstack) -- string_grouper_amd/distributed.py.  The tail of fit() (match
    list, groups) then runs on every rank's GPU over the whole result, as on one GPU."""
    name = "hip-distributed"

    def __init__(self, ctx: Optional[N.Context] = None, group=None):
        super().__init__(ctx)
        self.group = group

    def tfidf(self, master, duplicates, ngram_size, regex, ignore_case, normalize_to_ascii, dtype):
        import torch.distributed as dist
        from . import distributed as D
        rank, world = dist.get_rank(self.group), dist.get_world_size(self.group)

        def factory():
            return HipTfidfVectorizer(ngram_size=ngram_size, regex=regex, ignore_case=ignore_case,
                                      normalize_to_ascii=normalize_to_ascii, dtype=dtype, ctx=self.ctx)
        ops = D.HipOps(self.ctx, factory)
        self.timings = {}
        t = time.perf_counter()
        probe = factory()
        ranges, blocks = [], []
        for series in ([master] if duplicates is None else [master, duplicates]):
            lo, hi = D.row_block(rank, world, len(series))
            ranges.append((lo, hi))
            blocks.append(probe.prepare(series.iloc[lo:hi] if hasattr(series, "iloc") else series[lo:hi]))
        t = self._tick("prepare_and_upload_s", t)
        try:
            vec, mats = D.sharded_tfidf(ops, blocks, self.group)
        except D.ShardedFitNotApplicable:
            # (raised on every rank alike.)  n-gram keys coded over the alphabet of the strings at hand -- ngram_size > 3,
            # characters kept by normalize_to_ascii=False -- mean something else on every rank: every rank vectorises the
            # WHOLE columns (it holds them: the public API runs the same script on the same Series everywhere), and only
            # the multiply is shared: this rank's rows on the left, the whole matrix, which is already here, on the right.
            return self._tfidf_replicated(master, duplicates, factory, ops, ranges, t)
        self._tick("vectorise_s", t)
        A = ShardedMatrix(mats[0], len(master), ranges[0], ops, self.group)
        B = A if duplicates is None else ShardedMatrix(mats[1], len(duplicates), ranges[1], ops, self.group)
        return A, B, vec

    def _tfidf_replicated(self, master, duplicates, factory, ops, ranges, t):
        vec = factory()
        sets = [vec.prepare(master)] + ([] if duplicates is None else [vec.prepare(duplicates)])
        vec.fit_prepared(sets)
        full = [vec.transform_prepared(s) for s in sets]
        self._tick("vectorise_s", t)
        out = []
        for csr, (lo, hi) in zip(full, ranges):
            m = ShardedMatrix(csr.row_block(lo, hi), csr.dims()[0], (lo, hi), ops, self.group)
            m._full = csr                      # no all-gather: the whole matrix is this rank's own work
            out.append(m)
        return out[0], (out[0] if duplicates is None else out[1]), vec

    def _topn_device(self, A, B, top_n: int, threshold: float) -> "N.TopN":
        if not isinstance(A, ShardedMatrix):
            # replicated left side: every rank does it all -- against the WHOLE right-hand side, not this rank's block of it
            if isinstance(B, ShardedMatrix):
                B = DeviceMatrix(B.full())
            return super()._topn_device(A, B, top_n, threshold)
        from . import distributed as D
        right = B.full() if isinstance(B, ShardedMatrix) else B.csr
        try:
            res_local = D.sharded_topn(A._ops, A.csr, right, top_n, threshold, self_join=B is A, group=self.group)
        except OverflowError:
            # the right-hand side does not fit one inverted index (on every rank alike: it is the same matrix): the local
            # rows against row blocks of it, merged on the device (K5), as on one GPU and as fit() expects of the engine
            # (string_grouper.py:397-413)
            res_local = HipEngine._topn_device(self, DeviceMatrix(A.csr), DeviceMatrix(right), top_n, threshold)
        cols, vals, counts = D.gather_topn(A._ops, res_local, self.group, on_device=True)
        res_local.free()
        return A._ops.topn_from_tensors(cols, vals, counts, B.shape[0])

    def rowwise_dot(self, A, B) -> np.ndarray:
        """Row-wise similarity: local rows on the device (K9), the ranks' pieces concatenated."""
        if not isinstance(A, ShardedMatrix):
            return super().rowwise_dot(A, B)
        import torch
        from . import distributed as D
        local = self.ctx.rowwise_dot(A.csr, B.csr)
        parts = D.all_gather_ragged(torch.from_numpy(np.ascontiguousarray(local)).to(A._ops.device), self.group)
        return torch.cat(parts).cpu().numpy()

    def topn_multiply_blocked(self, A, B, n_blocks, top_n, threshold):
        if isinstance(A, ShardedMatrix):          # explicit n_blocks only cut the work differently: same result
            return self.topn_multiply(A, B, top_n, threshold).astype(np.float64)
        return super().topn_multiply_blocked(A, B, n_blocks, top_n, threshold)


def enable_distributed(group=None, ctx: Optional[N.Context] = None) -> "HipEngine":
    """Make fit() of this process use all ranks of the (already initialised) ``torch.distributed`` process group.
    Call on every rank after ``dist.init_process_group("nccl")`` and ``torch.cuda.set_device(local_rank)``."""
    import torch.distributed as dist
    N.lib()
    eng = DistributedHipEngine(ctx, group) if dist.is_initialized() and dist.get_world_size(group) > 1 else HipEngine(ctx)
    set_engine(eng)
    return eng


_engine = None


def get_engine():
    """The process-wide engine.  Raises (ImportError / RuntimeError) when libsg_hip.so is missing or
    no MI355X is visible -- the package never substitutes a CPU implementation."""
    global _engine
    if _engine is None:
        N.lib()
        _engine = HipEngine()
    return _engine


def set_engine(engine) -> None:
    """Test hook: inject an engine double (tests/_oracle_engine.py).  Not used by the product."""
    global _engine
    _engine = engine
