"""CPU restatement of the reference's ``match_strings`` END TO END -- TEST INFRASTRUCTURE / CPU BASELINE ONLY.

bench.py times this on the GPU box's host cores (``cpu_baseline``): the reference package itself is Python
under /root/reference, which does not exist on the GPU box, so its call sequence is restated here step by step
on top of the same third-party arithmetic the reference calls -- sklearn's TfidfVectorizer with the reference's
analyzer, scipy's lil round trip, pandas -- and ``oracle/sdtn_port.c`` for the absent ``sparse_dot_topn`` wheel.
``oracle/validate_ref_pipeline.py`` (run where /root/reference is mounted; log under profiles/) shows that this
restatement returns the unmodified reference's frames and takes the unmodified reference's time.

Reference lines followed (paths relative to /root/reference/string_grouper/string_grouper.py):
  match_strings                         :130-153
  StringGrouper.__init__/_build_corpus  :224-308   (vectoriser constructed AND fitted: tokenisation pass 1)
  fit                                   :380-431   (_get_tf_idf_matrices :685-697 -> _fit_vectorizer :699-707 =
                                                    pass 2, transform = pass 3; block guess :387-389;
                                                    _build_matches :709-752; lil / diagonal / symmetrise :417-427)
  _get_matches_list                     :755-763
  get_matches                           :442-500
Only the self-join / two-series cases without ids are restated (what the benchmark configurations use).
"""
from __future__ import annotations

import time
from typing import Optional

import numpy as np
import pandas as pd
import scipy.sparse as sp

from . import oracle as O
from . import port as P


def _vectorizer(dtype, **ngram_kw):
    from sklearn.feature_extraction.text import TfidfVectorizer
    return TfidfVectorizer(min_df=1, analyzer=lambda s: O.ngrams(s, **ngram_kw), dtype=dtype)   # :306


def build_matches_blocked(A, B, n_blocks, top_n, threshold, n_threads, clock=None):
    """:733-752 with the C port standing in for sp_matmul_topn and the oracle's zip."""
    As = [A[list(r)] for r in O.define_chunks(A.shape[0], n_blocks[0])]
    Bs = [B[list(r)] for r in O.define_chunks(B.shape[0], n_blocks[1])]
    Cs = [[P.sp_matmul_topn_port(Aj, Bi.T, top_n, threshold, True, n_threads) for Bi in Bs] for Aj in As]
    if clock is not None:
        clock("products")
    Czip = [zip_port(top_n, Cis) for Cis in Cs]
    if clock is not None:
        clock("zip")
    return sp.vstack(Czip, dtype=np.float64)


def zip_port(top_n, C_mats):
    """zip_sp_matmul_topn (:746) -- vectorised numpy (the per-row Python loop of oracle.zip_sp_matmul_topn would
    dominate a timing): stack the column blocks, order every row by (score desc, column asc), keep top_n."""
    if len(C_mats) == 1:
        return sp.csr_matrix(C_mats[0])
    M = sp.hstack(C_mats, format="csr")
    rows = np.repeat(np.arange(M.shape[0]), np.diff(M.indptr))
    order = np.lexsort((M.indices, -M.data, rows))
    r, c, v = rows[order], M.indices[order], M.data[order]
    first = np.searchsorted(r, np.arange(M.shape[0]))
    rank = np.arange(len(r)) - first[r]
    keep = rank < top_n
    r, c, v = r[keep], c[keep], v[keep]
    indptr = np.zeros(M.shape[0] + 1, dtype=np.int64)
    np.cumsum(np.bincount(r, minlength=M.shape[0]), out=indptr[1:])
    return sp.csr_matrix((v, c, indptr), shape=M.shape)


def match_strings_cpu(master: pd.Series, duplicates: Optional[pd.Series] = None, max_n_matches: int = 20,
                      min_similarity: float = 0.8, tfidf_matrix_dtype=np.float64, number_of_processes: int = 4,
                      n_blocks=None, timings: Optional[dict] = None, **ngram_kw) -> pd.DataFrame:
    t_last = [time.perf_counter()]

    def clock(name):
        now = time.perf_counter()
        if timings is not None:
            timings[name] = timings.get(name, 0.0) + now - t_last[0]
        t_last[0] = now

    strings = master if duplicates is None else pd.concat([master, duplicates])
    vec = _vectorizer(tfidf_matrix_dtype, **ngram_kw)
    vec.fit(strings)                                   # constructor: _build_corpus (:305-308)
    clock("vectorise_pass1_ctor_fit")
    vec.fit(strings)                                   # fit(): _fit_vectorizer again (:687, :699-707)
    clock("vectorise_pass2_fit")
    A = vec.transform(master)                          # :689
    B = A if duplicates is None else vec.transform(duplicates)
    clock("vectorise_pass3_transform")
    guess = O.guess_n_blocks(A.shape[0], B.shape[0])   # :387-389
    if n_blocks is None:
        n_blocks = guess
    C = build_matches_blocked(A, B, n_blocks, max_n_matches, min_similarity, number_of_processes, clock)
    clock("vstack")
    if duplicates is None:                             # force_symmetries default True (:417-427)
        C = C.tolil()
        r = np.arange(C.shape[0])
        C[r, r] = 1                                    # _fix_diagonal (:954-958)
        r, c = C.nonzero()
        C[c, r] = C[r, c]                              # _symmetrize_matrix (:960-964)
        C = C.tocsr()
        clock("lil_diagonal_symmetrise")
    else:
        C = C.tocsr()
    r, c = C.nonzero()                                 # _get_matches_list (:755-763)
    ml = pd.DataFrame({'master_side': r.astype(np.int64), 'dupe_side': c.astype(np.int64), 'similarity': C.data})
    clock("matches_list")
    # get_matches (:442-500), default index handling (ignore_index False)
    left = (master if master.name else master.rename('side')).iloc[ml.master_side].reset_index(drop=False)
    rsrc = master if duplicates is None else duplicates
    right = (rsrc if rsrc.name else rsrc.rename('side')).iloc[ml.dupe_side].reset_index(drop=False)
    right = right[right.columns[::-1]]
    out = pd.concat([left.rename(columns={x: f"left_{x}" for x in left.columns}), ml.similarity.reset_index(drop=True),
                     right.rename(columns={x: f"right_{x}" for x in right.columns})], axis=1)
    clock("get_matches_frames")
    return out
