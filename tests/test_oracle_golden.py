"""CPU tests: the oracle against the reference's golden vectors, its two TF-IDF restatements
against each other, and the C port against the scipy oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import oracle as O
from oracle import port as P
from tests import _golden as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ngrams_known_answers():
    ka = G.KNOWN
    assert O.ngrams("McDonalds", ignore_case=False) == ka["ngrams_McDonalds_case"]["value"]
    assert O.ngrams("McDonalds") == ka["ngrams_McDonalds_lower"]["value"]
    assert O.ngrams(ka["ngrams_unicode"]["input"]) == ka["ngrams_unicode"]["value"]
    assert O.ngrams("ab") == [] and O.ngrams("") == []
    assert O.ngrams("a.b,c-d/e f") == ["abc", "bcd", "cde", "def"]


@pytest.mark.parametrize("case", ["accounts_f64", "accounts_f32", "customers_f64", "customers_case", "customers_ngram2"])
def test_tfidf_matches_reference_fixture(case):
    g = G.CASES["tfidf_" + case]
    strings = G.INPUTS[g["input"]]
    kw = {k: v for k, v in G.kwargs_from_golden(g["kwargs"]).items()}
    dtype = kw.pop("tfidf_matrix_dtype", np.float64)
    expect = G.csr_from_golden(g["matrix"])
    for impl in (O.tfidf_sklearn, O.tfidf_numpy):
        (m,), vocab, idf = impl(strings, [strings], dtype=dtype, **kw)
        G.assert_csr_bitequal(m, expect, f"{case} {impl.__name__}")
        assert sorted(vocab, key=vocab.get) == g["vocabulary"]


def test_tfidf_master_and_duplicates_fixture():
    g = G.CASES["tfidf_customers_vs_customers2"]
    m, d = G.INPUTS["customers"], G.INPUTS["customers2"]
    for impl in (O.tfidf_sklearn, O.tfidf_numpy):
        (a, b), _, _ = impl(m + d, [m, d], dtype=np.float64)
        G.assert_csr_bitequal(a, G.csr_from_golden(g["master"]))
        G.assert_csr_bitequal(b, G.csr_from_golden(g["duplicates"]))


def test_tfidf_restatements_agree_on_synthetic_names():
    from string_grouper_amd.synth import synth_names
    names = synth_names(3000, 5) + ["", "ab", "x" * 200, "Ünïcödé Straße"]
    for dtype in (np.float32, np.float64):
        (a,), va, ia = O.tfidf_sklearn(names, [names], dtype=dtype)
        (b,), vb, ib = O.tfidf_numpy(names, [names], dtype=dtype)
        assert va == vb
        np.testing.assert_array_equal(ia, ib)
        G.assert_csr_bitequal(a, b)


def test_known_answer_matrices():
    ka = G.KNOWN
    (m,), _, _ = O.tfidf_sklearn(ka["tfidf_foo_bar_baz"]["input"], [ka["tfidf_foo_bar_baz"]["input"]])
    np.testing.assert_array_equal(m.toarray(), np.array(ka["tfidf_foo_bar_baz"]["dense"]))
    ms, ds = ka["tfidf_master_dupes"]["master"], ka["tfidf_master_dupes"]["dupes"]
    (a, b), _, _ = O.tfidf_sklearn(ms + ds, [ms, ds])
    np.testing.assert_array_equal(a.toarray(), np.array(ka["tfidf_master_dupes"]["master_dense"]))
    np.testing.assert_array_equal(b.toarray(), np.array(ka["tfidf_master_dupes"]["dupes_dense"]))
    C = O.build_matches(a, b, None, 20, 0.8)
    np.testing.assert_array_equal(C.toarray(), np.array(ka["build_matches_3x3"]["dense"]))
    # 0.08170638 (test:46-56): 'Hyper Startup Incorporated' x 'whatever' at min_similarity 0
    m_, d_ = ka["zero_min_similarity"]["master"], ka["zero_min_similarity"]["dupes"]
    (a, b), _, _ = O.tfidf_sklearn(m_ + d_, [m_, d_])
    C = O.sp_matmul_topn(a, b.T, 20, 0.0, True)
    assert C.nnz == 1 and abs(C[1, 0] - ka["zero_min_similarity"]["score_row1"]) < 5e-9


def _random_pair(seed, dtype, n=400, m=300, v=120):
    A = sp.random(n, v, density=0.08, random_state=seed, format="csr", dtype=np.float64)
    B = sp.random(m, v, density=0.08, random_state=seed + 1, format="csr", dtype=np.float64)
    A.data, B.data = np.abs(A.data) + 0.01, np.abs(B.data) + 0.01
    return A.astype(dtype), B.astype(dtype)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_port_equals_scipy_oracle(dtype):
    A, B = _random_pair(11, dtype)
    for top_n, thr, sort in ((5, 0.0, True), (3, 0.2, True), (1000, 0.05, True), (4, 0.1, False)):
        G.assert_csr_bitequal(P.sp_matmul_topn_port(A, B.T, top_n, thr, sort, 3), O.sp_matmul_topn(A, B.T, top_n, thr, sort))
        C1, C2 = P.sp_matmul_topn_port(A, B.T, top_n, thr, sort, 3), O.sp_matmul_topn(A, B.T, top_n, thr, sort)
        np.testing.assert_array_equal(C1.indices, C2.indices)      # within-row order too


def test_blocked_build_matches_equals_unblocked():
    """The reference's own invariant (test:191-336): every n_blocks gives the same matches."""
    A, B = _random_pair(3, np.float64)
    ref = O.build_matches(A, B, (1, 1), 7, 0.1)
    for nb in ((1, 2), (2, 1), (3, 2), (1, 8), (4, 4)):
        G.assert_csr_bitequal(O.build_matches(A, B, nb, 7, 0.1), ref, str(nb))


def test_threshold_is_strict_and_ties_are_canonical():
    A = sp.csr_matrix(np.array([[1.0, 0.0], [0.5, 0.0]]))
    B = sp.csr_matrix(np.array([[1.0, 0.0], [1.0, 0.0], [1.0, 0.0], [0.5, 0.0]]))
    C = O.sp_matmul_topn(A, B.T, 2, 0.5, True)
    assert C[0].indices.tolist() == [0, 1]          # three-way tie at 1.0 -> lowest columns
    assert C[1].nnz == 0                            # 0.5 is not > 0.5
    G.assert_csr_bitequal(C, P.sp_matmul_topn_port(A, B.T, 2, 0.5, True, 1))


@pytest.mark.skipif(not os.path.isdir("/root/reference/string_grouper"), reason="reference tree not mounted")
@pytest.mark.parametrize("backend", ["oracle", "port"])
def test_unmodified_reference_suite_passes_on_the_oracle(backend):
    """Pins the oracle: the reference's own 53 unit tests run against the unmodified reference
    package with the oracle standing in for the absent sparse_dot_topn wheel."""
    env = dict(os.environ, SG_SHIM_BACKEND=backend,
               PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "tests", "ref_shims"), "/root/reference"]))
    r = subprocess.run([sys.executable, "-m", "pytest", "/root/reference/string_grouper/test", "-q", "-p",
                        "no:cacheprovider"], cwd="/tmp", env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "53 passed" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
