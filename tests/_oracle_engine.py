"""Engine double for CPU-only tests of the host logic: same four operations as
string_grouper_amd.engine.HipEngine, computed by the oracle.  TEST INFRASTRUCTURE ONLY -- the
product never imports this (string_grouper_amd.engine.get_engine() has no CPU option)."""
import numpy as np
import scipy.sparse as sp

from oracle import oracle as O
from oracle import port as P


class HostMatrix:
    def __init__(self, m):
        self.m = sp.csr_matrix(m)
        self.shape = self.m.shape
        self.dtype = self.m.dtype
        self.nnz = self.m.nnz

    def to_scipy(self):
        return self.m


class OracleEngine:
    name = "oracle"

    def __init__(self, use_port=False):
        self.use_port = use_port
        self.calls = []

    def tfidf(self, master, duplicates, ngram_size, regex, ignore_case, normalize_to_ascii, dtype):
        kw = dict(ngram_size=ngram_size, regex=regex, ignore_case=ignore_case, normalize_to_ascii=normalize_to_ascii)
        fit = list(master) + (list(duplicates) if duplicates is not None else [])
        sets = [list(master)] + ([list(duplicates)] if duplicates is not None else [])
        mats, vocab, idf = O.tfidf_sklearn(fit, sets, dtype=dtype, **kw)
        A = HostMatrix(mats[0])
        B = A if duplicates is None else HostMatrix(mats[1])
        return A, B, {"vocabulary_": vocab, "idf_": idf}

    def wrap(self, m):
        return m if isinstance(m, HostMatrix) else HostMatrix(m)

    def _mul(self, A, B, top_n, thr):
        thr = max(float(thr), 0.0)
        if self.use_port:
            return P.sp_matmul_topn_port(A, B.T, top_n, thr, True, 4)
        return O.sp_matmul_topn(A, B.T, top_n, thr, True)

    def topn_multiply(self, A, B, top_n, threshold):
        self.calls.append(("single", A.shape, B.shape))
        return self._mul(A.m, B.m, top_n, threshold)

    def topn_multiply_blocked(self, A, B, n_blocks, top_n, threshold):
        self.calls.append(("blocked", A.shape, B.shape, tuple(n_blocks)))
        As = [A.m[list(r)] for r in O.define_chunks(A.shape[0], n_blocks[0])]
        Bs = [B.m[list(r)] for r in O.define_chunks(B.shape[0], n_blocks[1])]
        Cs = [[self._mul(Aj, Bi, top_n, threshold) for Bi in Bs] for Aj in As]
        return sp.vstack([O.zip_sp_matmul_topn(top_n, row) for row in Cs], dtype=np.float64).tocsr()
