"""GPU parity tests proper: the HIP path (through the C ABI) against the oracle on the same
seeded inputs.  Integer / index results must be bit-exact; scores are compared bit-exact too
(the kernels reproduce the oracle's operation order), which implies the 1e-6 fp32 tolerance
BASELINE.json states."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import oracle as O
from oracle import port as P

pytestmark = pytest.mark.gpu


def _names(n, seed=1234):
    from string_grouper_amd.synth import synth_names
    return synth_names(n, seed)


def _tfidf(names, dtype):
    (m,), vocab, idf = O.tfidf_sklearn(names, [names], dtype=dtype)
    return m


def assert_csr_identical(C_dev, C_ref, what=""):
    assert C_dev.shape == C_ref.shape, what
    np.testing.assert_array_equal(np.asarray(C_dev.indptr, np.int64), np.asarray(C_ref.indptr, np.int64), err_msg=what)
    np.testing.assert_array_equal(C_dev.indices, C_ref.indices, err_msg=what)
    assert C_dev.data.dtype == C_ref.data.dtype, what
    np.testing.assert_array_equal(C_dev.data, C_ref.data, err_msg=what)      # bit-exact scores


@pytest.fixture(scope="module")
def mats():
    names = _names(20000)
    return {np.float32: _tfidf(names, np.float32), np.float64: _tfidf(names, np.float64)}


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("top_n,thr,sort", [(10, 0.8, True), (1, 0.8, True), (3, 0.5, True), (5, 0.3, False),
                                            (40, 0.0, True), (64, 0.1, True), (65, 0.1, True), (130, 0.05, True)])
def test_spgemm_topn_selfjoin_matches_oracle(ctx, mats, dtype, top_n, thr, sort):
    from string_grouper_amd.sparse_dot_topn import sp_matmul_topn
    A = mats[dtype]
    C_dev = sp_matmul_topn(A, A.T, top_n, thr, sort=sort, ctx=ctx)
    C_ref = P.sp_matmul_topn_port(A, A.T, top_n, thr, sort, 8)
    assert_csr_identical(C_dev, C_ref, f"{dtype.__name__} top_n={top_n} thr={thr} sort={sort}")


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_top_n_above_the_register_list_takes_the_pruned_kernel_and_hands_full_rows_on(ctx, dtype, monkeypatch):
    """max_n_matches of 65 .. 128 at a name-matching threshold: the pruned kernel keeps a row's best 64; rows whose list
    comes out full (hubs of more than 64 near-identical names) are redone by the exact kernel, a pass per 64 entries.
    Self-join and master x duplicates, identical rows grouped or not -- the port's result, bit for bit."""
    from string_grouper_amd.sparse_dot_topn import sp_matmul_topn
    rng = np.random.default_rng(8)
    names = list(_names(15000, seed=3))
    for hub, size in (("NORTHERN LIGHTS HOLDING CO", 150), ("BLUE RIVER PARTNERS", 70), ("KAPPA LTD", 64)):
        for k, at in enumerate(rng.choice(len(names), size, replace=False)):
            names[at] = hub + (" " + "ABCDEFGH"[k % 8] if k % 3 == 0 else "")      # identical and near-identical members
    A = _tfidf(names, dtype)
    for collapse in ("0", "1"):
        monkeypatch.setenv("SG_COLLAPSE", collapse)
        for top_n, thr in ((100, 0.8), (65, 0.6), (128, 0.7)):
            got = sp_matmul_topn(A, A.T, top_n, thr, sort=True, ctx=ctx)
            st = ctx.stats()
            assert_csr_identical(got, P.sp_matmul_topn_port(A, A.T, top_n, thr, True, 8), f"top_n={top_n} thr={thr} collapse={collapse}")
            if collapse == "0":
                assert st["prune_rows"] > 14000 and 60 < st["exact_rows"] < 2000, st      # pruned, hub rows handed on
        left = A[2000:9000]
        assert_csr_identical(sp_matmul_topn(left, A.T, 90, 0.75, sort=False, ctx=ctx), P.sp_matmul_topn_port(left, A.T, 90, 0.75, False, 8),
                             f"one-sided, sorted by column, collapse={collapse}")
    # 129 and more: the exact kernel, as before
    monkeypatch.setenv("SG_COLLAPSE", "0")
    got = sp_matmul_topn(A, A.T, 129, 0.8, sort=True, ctx=ctx)
    assert ctx.stats()["prune_rows"] == 0
    assert_csr_identical(got, P.sp_matmul_topn_port(A, A.T, 129, 0.8, True, 8), "top_n=129")


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_top_n_above_the_register_list_inside_the_selfjoin_form(ctx, dtype, monkeypatch):
    """Round 6: max_n_matches of 65 .. 128 no longer leaves the self-join form.  The pass sends a row's own matches (j < i)
    through the pair list as well -- "row i receives column j" -- and the second pass selects with two register lists
    (128 entries); no row goes to the exact kernel for its top_n.  Hubs of 150 / 70 / 64 identical and near-identical
    names (rows with more matches than 64, than top_n, ties at the cut), long names in the wide launch and beyond it (the
    exact kernel's self-join launch inside the pass), identical rows grouped or not -- the port's result, bit for bit."""
    from string_grouper_amd.sparse_dot_topn import sp_matmul_topn
    rng = np.random.default_rng(9)
    names = list(_names(24000, seed=5))
    for hub, size in (("NORTHERN LIGHTS HOLDING CO", 150), ("BLUE RIVER PARTNERS", 70), ("KAPPA LTD", 64), ("OMEGA TRADING", 129)):
        for k, at in enumerate(rng.choice(len(names), size, replace=False)):
            names[at] = hub + (" " + "ABCDEFGH"[k % 8] if k % 3 == 0 else "")
    for at in rng.choice(len(names), 40, replace=False):        # rows of 65 .. 128 non-zeros, and a few beyond
        names[at] = " ".join(names[(at + q) % len(names)] for q in range(4 if at % 2 else 9))
    A = _tfidf(names, dtype)
    monkeypatch.setenv("SG_SYM", "1")                          # (the form starts at 65 536 rows by itself)
    for collapse in ("0", "1"):
        monkeypatch.setenv("SG_COLLAPSE", collapse)
        for top_n, thr in ((100, 0.8), (65, 0.6), (128, 0.7), (127, 0.5)):
            got = sp_matmul_topn(A, A.T, top_n, thr, sort=True, ctx=ctx)
            st = ctx.stats()
            assert_csr_identical(got, P.sp_matmul_topn_port(A, A.T, top_n, thr, True, 8), f"top_n={top_n} thr={thr} collapse={collapse}")
            assert st["prune_symmetric"] == 1 and st["prune_rows"] > 15000, st
            assert st["exact_rows"] < 40, st                   # only the rows beyond 128 non-zeros: nobody for a full list
    monkeypatch.setenv("SG_COLLAPSE", "0")
    got = sp_matmul_topn(A, A.T, 100, 0.8, sort=False, ctx=ctx)
    assert_csr_identical(got, P.sp_matmul_topn_port(A, A.T, 100, 0.8, False, 8), "sorted by column")
    monkeypatch.setenv("SG_SYM_PAIR_CAP", "2000")              # a pair list that is too small: the one-sided form, as before
    got = sp_matmul_topn(A, A.T, 100, 0.8, sort=True, ctx=ctx)
    assert ctx.stats()["prune_symmetric"] == 0
    assert_csr_identical(got, P.sp_matmul_topn_port(A, A.T, 100, 0.8, True, 8), "fallback")


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_thresholds_below_the_pruned_kernels_run_the_exact_kernel_in_the_selfjoin_form(ctx, dtype, monkeypatch):
    """Round 6: below 0.45 the prefix filter passes too much and the exact kernel takes the product -- now in the self-join
    form: every row through the exact kernel's self-join launch (pairs j <= i over the tiles up to its own, mirrored pairs
    through the pair list), half the (row, tile) visits.  Same bits as the one-sided exact kernel and the port; a pair
    list that runs full sends the product back to the one-sided form."""
    from string_grouper_amd.sparse_dot_topn import sp_matmul_topn
    rng = np.random.default_rng(10)
    names = list(_names(24000, seed=6))
    for k, at in enumerate(rng.choice(len(names), 90, replace=False)):
        names[at] = "HARBOUR VIEW ESTATES" + (" " + "ABCDEFGH"[k % 8] if k % 3 == 0 else "")
    A = _tfidf(names, dtype)
    monkeypatch.setenv("SG_SYM", "1")
    for collapse in ("0", "1"):
        monkeypatch.setenv("SG_COLLAPSE", collapse)
        for top_n, thr in ((10, 0.39), (20, 0.3), (100, 0.35), (3, 0.1), (64, 0.2)):
            got = sp_matmul_topn(A, A.T, top_n, thr, sort=True, ctx=ctx)
            st = ctx.stats()
            assert_csr_identical(got, P.sp_matmul_topn_port(A, A.T, top_n, thr, True, 8), f"top_n={top_n} thr={thr} collapse={collapse}")
            assert st["prune_rows"] == 0, st
            if thr >= 0.3:                                      # (at 0.1 the pair list may run full: one-sided, same result)
                assert st["prune_symmetric"] == 1 and st["exact_rows"] > 15000, st
    # large matrices send the most expensive rows first and judge the pair list by what they write (from a million rows on;
    # here through the test hook): two launches, the same bits -- and a list the estimate says is too small calls the form off
    monkeypatch.setenv("SG_COLLAPSE", "0")
    monkeypatch.setenv("SG_EXACT_SYM_PILOT_ROWS", "700")
    got = sp_matmul_topn(A, A.T, 10, 0.35, sort=True, ctx=ctx)
    assert ctx.stats()["prune_symmetric"] == 1 and ctx.stats()["exact_rows"] == A.shape[0]
    assert_csr_identical(got, P.sp_matmul_topn_port(A, A.T, 10, 0.35, True, 8), "rows in two launches")
    monkeypatch.setenv("SG_SYM_PAIR_CAP", "40000")
    got = sp_matmul_topn(A, A.T, 10, 0.35, sort=True, ctx=ctx)
    assert ctx.stats()["prune_symmetric"] == 0
    assert_csr_identical(got, P.sp_matmul_topn_port(A, A.T, 10, 0.35, True, 8), "called off after the first rows")
    monkeypatch.delenv("SG_SYM_PAIR_CAP")
    monkeypatch.delenv("SG_EXACT_SYM_PILOT_ROWS")
    monkeypatch.setenv("SG_EXACT_SYM", "0")                    # the switch: one-sided exact kernel as in round 5
    got = sp_matmul_topn(A, A.T, 10, 0.35, sort=True, ctx=ctx)
    assert ctx.stats()["prune_symmetric"] == 0
    assert_csr_identical(got, P.sp_matmul_topn_port(A, A.T, 10, 0.35, True, 8), "SG_EXACT_SYM=0")


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_selfjoins_below_name_matching_thresholds_take_the_tile_by_tile_form(ctx, dtype, monkeypatch):
    """Round 6: from 0.65 down a self-join runs the pruned multiply's tile-by-tile form on an index of its own (2048-column
    tiles, an accumulator per column; built on first use, kept with the index) -- the stream form's folded accumulators
    raise ever more false alarms as the threshold falls (scripts/form_sweep.py) -- and the pruned multiply's envelope
    then starts at 0.40.  Self-join form or one-sided, identical rows grouped or not, top_n on both sides of the register
    list: the port's bits; and the same bits as with the switch off."""
    from string_grouper_amd.vectorizer import HipTfidfVectorizer
    rng = np.random.default_rng(11)
    names = list(_names(30000, seed=12))
    for k, at in enumerate(rng.choice(len(names), 80, replace=False)):
        names[at] = "SILVER LAKE CAPITAL" + (" " + "ABCDEFGH"[k % 8] if k % 3 == 0 else "")
    (A_ref,), _, _ = O.tfidf_sklearn(names, [names], dtype=dtype)
    vec = HipTfidfVectorizer(dtype=dtype, ctx=ctx)
    p = vec.prepare(names)
    vec.fit_prepared([p])
    dA = vec.transform_prepared(p)
    for collapse in ("1", "0"):
        monkeypatch.setenv("SG_COLLAPSE", collapse)
        post = ctx.postings_build(dA)
        for sym in ("1", "0"):
            monkeypatch.setenv("SG_SYM", sym)
            for top_n, thr in ((10, 0.4), (20, 0.5), (10, 0.64), (100, 0.45), (64, 0.6)):
                res = ctx.spgemm_topn(dA, post, top_n, thr, True)
                st = ctx.stats()
                what = f"{dtype.__name__} top_n={top_n} thr={thr} SG_SYM={sym} SG_COLLAPSE={collapse}"
                assert_csr_identical(res.to_scipy(), P.sp_matmul_topn_port(A_ref, A_ref.T, top_n, thr, True, 8), what)
                res.free()
                assert st["prune_rows"] > 0 and st["prune_symmetric"] == int(sym), (what, st)
        # ... the same index serves the name-matching thresholds in the stream form, before and after
        res = ctx.spgemm_topn(dA, post, 10, 0.8, True)
        assert_csr_identical(res.to_scipy(), P.sp_matmul_topn_port(A_ref, A_ref.T, 10, 0.8, True, 8), "0.8 on the same index")
        res.free()
        # large matrices send the last positions first and judge the pair list by what they write (a million rows and more in
        # the tile-by-tile form; here through the test hook, in both forms): two passes, the same bits -- and a list the
        # estimate says is too small calls the form off before the rest is multiplied
        monkeypatch.setenv("SG_SYM", "1")
        monkeypatch.setenv("SG_SYM_PILOT_ROWS", "900")
        for top_n, thr in ((20, 0.5), (10, 0.8), (100, 0.6)):
            res = ctx.spgemm_topn(dA, post, top_n, thr, True)
            assert ctx.stats()["prune_symmetric"] == 1
            assert_csr_identical(res.to_scipy(), P.sp_matmul_topn_port(A_ref, A_ref.T, top_n, thr, True, 8), f"first rows first, top {top_n} at {thr}")
            res.free()
        monkeypatch.setenv("SG_SYM_PAIR_CAP", "30000")
        res = ctx.spgemm_topn(dA, post, 20, 0.5, True)
        assert ctx.stats()["prune_symmetric"] == 0
        assert_csr_identical(res.to_scipy(), P.sp_matmul_topn_port(A_ref, A_ref.T, 20, 0.5, True, 8), "called off after the first rows")
        res.free()
        monkeypatch.delenv("SG_SYM_PAIR_CAP")
        monkeypatch.delenv("SG_SYM_PILOT_ROWS")
        monkeypatch.delenv("SG_SYM")
        monkeypatch.setenv("SG_ALT_FORM", "0")                 # the switch: the stream form at any threshold
        res = ctx.spgemm_topn(dA, post, 20, 0.5, True)
        assert_csr_identical(res.to_scipy(), P.sp_matmul_topn_port(A_ref, A_ref.T, 20, 0.5, True, 8), "SG_ALT_FORM=0")
        res.free()
        monkeypatch.delenv("SG_ALT_FORM")
        post.free()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_long_strings_on_both_sides_of_the_pruned_kernels_row_length(ctx, dtype):
    """Round 6: strings of ~100 n-grams (the wide launch, now sized by the rows' average length, and the self-join form from
    19 x 65 536 non-zeros on) and of ~155 (beyond 128 entries a row ON AVERAGE the pruned multiply would pass nearly every
    row on one by one: the exact kernel takes the product, in the self-join form from 16 384 rows).  The port's bits."""
    from string_grouper_amd.sparse_dot_topn import sp_matmul_topn
    base = _names(8 * 17000, seed=21)
    for k, n, pruned in ((5, 14000, True), (8, 17000, False)):
        names = [" ".join(base[k * i:k * i + k]) for i in range(n)]
        names[100:130] = [names[7]] * 30                        # a hub of identical long rows
        A = _tfidf(names, dtype)
        assert (A.nnz > 128 * n) != pruned
        got = sp_matmul_topn(A, A.T, 10, 0.8, sort=True, ctx=ctx)
        st = ctx.stats()
        assert_csr_identical(got, P.sp_matmul_topn_port(A, A.T, 10, 0.8, True, 16), f"{k} names joined, {dtype.__name__}")
        assert st["prune_symmetric"] == 1 and (st["prune_rows"] > 0) == pruned, st
        left = A[1000:5000]                                       # one-sided: the wide launch / the exact kernel's row list
        assert_csr_identical(sp_matmul_topn(left, A.T, 10, 0.8, sort=True, ctx=ctx), P.sp_matmul_topn_port(left, A.T, 10, 0.8, True, 16),
                             f"{k} names joined, one-sided")


def test_random_lists_with_repeats_equal_the_port(ctx):
    """Seeded random jobs around the switches of round 3 -- size on both sides of the grouping's threshold, share and size of
    the repeats, near-duplicates of the hubs (ties at the cut between a group and single rows), top_n from 1 to 128,
    thresholds from 0.5 to 0.95, both dtypes, sorted or shuffled, self-join (as the library sees it: the same device
    matrix on both sides, so the self-join form can run) and one-sided -- against the port, bit for bit."""
    import os
    rng = np.random.default_rng(int(os.environ.get("SG_TEST_RANDOM_SEED", "2024")))
    for job in range(int(os.environ.get("SG_TEST_RANDOM_JOBS", "24"))):     # (a longer soak: SG_TEST_RANDOM_JOBS=200)
        n = int(rng.choice([6000, 9000, 14000, 30000, 70000]))
        names = list(_names(n, seed=100 + job))
        for _ in range(int(rng.integers(0, 6))):                         # hubs
            hub = names[int(rng.integers(0, n))]
            size = int(rng.choice([3, 20, 64, 65, 200, 1500]))
            for k, at in enumerate(rng.choice(n, min(size, n // 4), replace=False)):
                names[at] = hub if k % 4 else hub + " " + "XYZW"[k % 3]
        share = float(rng.choice([0.0, 0.02, 0.05, 0.3]))                # plain repeats
        for at in rng.choice(n - 1, int(share * n), replace=False):
            names[at + 1] = names[at]
        if rng.random() < 0.5:
            names = sorted(names)
        dtype = np.float32 if rng.random() < 0.6 else np.float64
        top_n = int(rng.choice([1, 2, 10, 10, 20, 63, 64, 65, 100, 128]))
        thr = float(rng.choice([float(x) for x in os.environ["SG_TEST_RANDOM_THRESHOLDS"].split(",")] if os.environ.get("SG_TEST_RANDOM_THRESHOLDS")
                               else [0.5, 0.6, 0.75, 0.8, 0.8, 0.9, 0.95]))      # (a soak of the low-threshold forms: e.g. 0.25,0.35,0.42,0.55,0.62)
        sym = str(rng.choice(["", "0", "1"]))
        what = f"job {job}: n={n} top_n={top_n} thr={thr} {dtype.__name__} SG_SYM={sym!r} repeats={share}"
        A = _tfidf(names, dtype)
        dA = ctx.csr_from_scipy(A)
        if sym:
            ctx.set_option("SG_SYM", sym)
        post = ctx.postings_build(dA)
        res = ctx.spgemm_topn(dA, post, top_n, thr, True)
        assert_csr_identical(res.to_scipy(), P.sp_matmul_topn_port(A, A.T, top_n, thr, True, 16), what)
        res.free()
        lo = int(rng.integers(0, n // 2))
        left = A[lo:lo + n // 3]
        dL = ctx.csr_from_scipy(left)
        res = ctx.spgemm_topn(dL, post, top_n, thr, True)
        assert_csr_identical(res.to_scipy(), P.sp_matmul_topn_port(left, A.T, top_n, thr, True, 16), what + " one-sided")
        for h in (res, dL, post, dA):
            h.free()
        ctx.reset_options()


@pytest.mark.parametrize("tile_cols", [1024, 2048, 4096, 8192])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_spgemm_tile_sizes_and_groups(ctx, mats, dtype, tile_cols, monkeypatch):
    A = mats[dtype][:5000]
    B = mats[dtype][3000:20000]
    C_ref = P.sp_matmul_topn_port(A, B.T, 10, 0.6, True, 8)
    dA, dB = ctx.csr_from_scipy(A), ctx.csr_from_scipy(B)
    post = ctx.postings_build(dB, tile_cols)
    for group in ("1", "3", "0"):
        monkeypatch.setenv("SG_TILE_GROUP", group)
        res = ctx.spgemm_topn(dA, post, 10, 0.6, True)
        assert_csr_identical(res.to_scipy(), C_ref, f"tile={tile_cols} group={group}")


def test_spgemm_against_scipy_oracle_small(ctx, mats):
    """The scipy-product oracle itself (not only the C port) at a size it finishes in seconds."""
    from string_grouper_amd.sparse_dot_topn import sp_matmul_topn
    A = mats[np.float32][:3000]
    B = mats[np.float32][:6000]
    C_dev = sp_matmul_topn(A, B.T, 7, 0.4, sort=True, ctx=ctx)
    C_ref = O.sp_matmul_topn(A, B.T, 7, 0.4, True)
    assert_csr_identical(C_dev, C_ref)


def test_spgemm_edge_cases(ctx):
    from string_grouper_amd.sparse_dot_topn import sp_matmul_topn
    rng = np.random.default_rng(7)
    # empty rows, rows longer than 64 non-zeros, more top_n than right rows, single right row
    A = sp.random(300, 500, density=0.05, random_state=3, dtype=np.float64, format="csr")
    A.data = np.abs(A.data) + 0.01
    dense_row = sp.csr_matrix(np.abs(rng.standard_normal((1, 500))) + 0.01)
    A = sp.vstack([A[:100], sp.csr_matrix((5, 500)), dense_row, A[100:]]).tocsr()
    B = sp.random(37, 500, density=0.3, random_state=5, dtype=np.float64, format="csr")
    B.data = np.abs(B.data) + 0.01
    for dtype in (np.float32, np.float64):
        Ad, Bd = A.astype(dtype), B.astype(dtype)
        for top_n, thr in ((5, 0.0), (100, 0.5), (37, 0.0), (1, 2.0)):
            C_dev = sp_matmul_topn(Ad, Bd.T, top_n, thr, sort=True, ctx=ctx)
            C_ref = P.sp_matmul_topn_port(Ad, Bd.T, top_n, thr, True, 2)
            assert_csr_identical(C_dev, C_ref, f"{dtype.__name__} top_n={top_n} thr={thr}")
        C_dev = sp_matmul_topn(Ad, Bd[:1].T, 3, 0.0, sort=True, ctx=ctx)
        assert_csr_identical(C_dev, P.sp_matmul_topn_port(Ad, Bd[:1].T, 3, 0.0, True, 1), "one right row")
    # all-empty left matrix
    Z = sp.csr_matrix((10, 500), dtype=np.float32)
    C_dev = sp_matmul_topn(Z, B.astype(np.float32).T, 3, 0.0, sort=True, ctx=ctx)
    assert C_dev.nnz == 0 and C_dev.shape == (10, 37)


def test_ties_at_the_cut_are_canonical(ctx):
    """Exact duplicates all score 1.0: the cut must keep the lowest columns (score desc, col asc)."""
    from string_grouper_amd.sparse_dot_topn import sp_matmul_topn
    names = ["ACME HOLDINGS INC"] * 40 + ["OTHER THING LLC"] * 3
    for dtype in (np.float32, np.float64):
        A = _tfidf(names, dtype)
        C_dev = sp_matmul_topn(A, A.T, 5, 0.8, sort=True, ctx=ctx)
        C_ref = P.sp_matmul_topn_port(A, A.T, 5, 0.8, True, 1)
        assert_csr_identical(C_dev, C_ref)
        assert list(C_dev[7].indices) == [0, 1, 2, 3, 4]


def test_zip_matches_oracle(ctx, mats):
    from string_grouper_amd.sparse_dot_topn import sp_matmul_topn, zip_sp_matmul_topn
    A = mats[np.float32][:4000]
    B = mats[np.float32][:9000]
    blocks = [B[0:2500], B[2500:2600], B[2600:9000]]
    Cs = [sp_matmul_topn(A, Bi.T, 6, 0.5, sort=True, ctx=ctx) for Bi in blocks]
    C_zip = zip_sp_matmul_topn(6, Cs, ctx=ctx)
    C_ref = O.zip_sp_matmul_topn(6, [P.sp_matmul_topn_port(A, Bi.T, 6, 0.5, True, 4) for Bi in blocks])
    assert_csr_identical(C_zip, C_ref)
    # and the zipped result equals the unblocked multiply (the reference's own invariant, test:191-336)
    C_one = sp_matmul_topn(A, B.T, 6, 0.5, sort=True, ctx=ctx)
    assert_csr_identical(C_zip, C_one)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_vectoriser_matches_sklearn_bitexact(ctx, dtype):
    from string_grouper_amd.vectorizer import HipTfidfVectorizer
    names = _names(30000, seed=99)
    names[17] = "ab"                 # shorter than an n-gram: empty row
    names[18] = ""                   # empty string
    names[19] = "A. B. C. Enterprises, Inc./-"  # punctuation removed by the regex
    names[20] = "x" * 300            # long row (> 64 n-grams, single distinct n-gram)
    names[21] = "".join(chr(97 + (i * 7) % 26) for i in range(500))
    (m_ref,), vocab, idf = O.tfidf_sklearn(names, [names], dtype=dtype)
    vec = HipTfidfVectorizer(dtype=dtype, ctx=ctx)
    m_dev = vec.fit(names).transform(names)
    assert vec.vocabulary_ == vocab
    np.testing.assert_array_equal(vec.idf_, idf)
    assert_csr_identical(m_dev, sp.csr_matrix(m_ref))


@pytest.mark.parametrize("env", [{"SG_DF_MARKS": "0"}, {"SG_K2_PLAIN": "1"}, {"SG_POSTINGS_SPLIT": "1"},
                                 {"SG_POSTINGS_SPLIT": "4"}, {"SG_POSTINGS_LDS": "0"}, {"SG_FILL_STAGED": "0"},
                                 {"SG_FILL_STAGE_CAP": "3000"}, {"SG_FILL_STAGE_CAP": "19500"},
                                 {"SG_POSTINGS_SPLIT": "1", "SG_FILL_STAGE_CAP": "19500"}])
def test_alternative_forms_of_k1_k2_k3_give_the_same_bits(ctx, env, monkeypatch):
    """Every kernel that got a faster form in round 2 keeps its first form behind a switch (document frequencies by
    global atomics -- the form the multi-GPU fit uses --, K2 with a thread per row, K3 without the tile split / without
    LDS counters; round 6: filter postings written one by one instead of staged in LDS, and chunks that do not fit the stage):
    each must still reproduce sklearn and the port bit for bit."""
    from string_grouper_amd.vectorizer import HipTfidfVectorizer
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    names = _names(20000, seed=17)
    names[5] = ""
    names[6] = "x" * 300
    names[7] = "".join(chr(97 + (i * 11) % 26) for i in range(400))
    for dtype in (np.float32, np.float64):
        (m_ref,), vocab, idf = O.tfidf_sklearn(names, [names], dtype=dtype)
        vec = HipTfidfVectorizer(dtype=dtype, ctx=ctx)
        p = vec.prepare(names)
        dA = vec.fit_prepared([p]).transform_prepared(p)
        assert vec.vocabulary_ == vocab
        np.testing.assert_array_equal(vec.idf_, idf)
        A = dA.to_scipy()
        assert_csr_identical(A, sp.csr_matrix(m_ref))
        post = ctx.postings_build(dA)
        res = ctx.spgemm_topn(dA, post, 10, 0.8, True)
        assert_csr_identical(res.to_scipy(), P.sp_matmul_topn_port(A, A.T, 10, 0.8, True, 8))
        res.free()
        post.free()
        dA.free()


def test_vectoriser_master_and_duplicates(ctx):
    """fit on master + duplicates, transform each (string_grouper.py:689-706); out-of-vocabulary
    n-grams of a third series are dropped like sklearn does."""
    from string_grouper_amd.vectorizer import HipTfidfVectorizer
    master = _names(5000, seed=1)
    dupes = _names(3000, seed=2)
    other = _names(1000, seed=3)
    (a_ref, b_ref, c_ref), vocab, idf = O.tfidf_sklearn(master + dupes, [master, dupes, other], dtype=np.float32)
    vec = HipTfidfVectorizer(dtype=np.float32, ctx=ctx)
    pm, pd_, po = vec.prepare(master), vec.prepare(dupes), vec.prepare(other)
    vec.fit_prepared([pm, pd_])
    assert vec.vocabulary_ == vocab
    assert_csr_identical(vec.transform_prepared(pm).to_scipy(), sp.csr_matrix(a_ref))
    assert_csr_identical(vec.transform_prepared(pd_).to_scipy(), sp.csr_matrix(b_ref))
    assert_csr_identical(vec.transform_prepared(po).to_scipy(), sp.csr_matrix(c_ref))


def test_vectoriser_unicode_case_and_ngram_sizes(ctx):
    from string_grouper_amd.vectorizer import HipTfidfVectorizer
    names = ["ÀbracâDABRÀ Straße GmbH", "McDonalds", "mcdonald's İstanbul", "Hyper-Startup Inc.", "ﬁne ﬂour Ⅻ", "naïve café"] * 3
    for kw in (dict(), dict(ignore_case=False), dict(ngram_size=2), dict(ngram_size=4), dict(ngram_size=5),
               dict(regex=r"[aeiou]"), dict(regex=r"inc\.?|\s")):
        (m_ref,), vocab, idf = O.tfidf_sklearn(names, [names], dtype=np.float64, **kw)
        vec = HipTfidfVectorizer(dtype=np.float64, ctx=ctx, **kw)
        m_dev = vec.fit(names).transform(names)
        assert vec.vocabulary_ == vocab, kw
        assert_csr_identical(m_dev, sp.csr_matrix(m_ref), str(kw))


UNICODE_NAMES = ["Ünïcödé Straße GmbH", "™ TRADEMARK Co", "№ 5 ℡ 12 ㎆", "İstanbul ISTANBUL ıi", "ΟΔΥΣΣΕΥΣ ΣΟΦΟΣ Σ", "ὈΔΥΣΣΕΎΣ",
                 "ﬁne ﬂour ﬃ Ⅻ ½", "ＦＵＬＬ　ｗｉｄｔｈ", "東京 Holdings 株式会社", "東京ホールディングス株式会社", "😀 emoji 𝔘𝔫𝔦", "ẞ ß SS",
                 "Crème Brûlée Holdings", "CREME BRULEE HOLDINGS", "Łódź Fabryka SA", "", "ab", "ACME Inc.", "acme, inc"]


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_vectoriser_covers_the_analyzers_whole_domain(ctx, dtype):
    """Every option combination of the analyzer (string_grouper.py:365-378) on names with non-ASCII characters:
    normalize_to_ascii on / off (off = n-grams over code points: symbol columns, the alphabet of the fit),
    ignore_case on / off, character-class and general regexes, several n-gram sizes -- vocabulary and matrix equal to
    sklearn driven with the reference's analyzer."""
    from string_grouper_amd.vectorizer import HipTfidfVectorizer
    names = UNICODE_NAMES * 2 + _names(300, seed=5)
    for norm in (True, False):
        for case in (True, False):
            for kw in (dict(), dict(ngram_size=2), dict(ngram_size=4), dict(regex=r"[aeiouéß]"), dict(regex=r"inc\.?|\s")):
                kw = dict(kw, normalize_to_ascii=norm, ignore_case=case)
                (m_ref,), vocab, idf = O.tfidf_sklearn(names, [names], dtype=dtype, **kw)
                vec = HipTfidfVectorizer(dtype=dtype, ctx=ctx, **kw)
                m_dev = vec.fit_transform(names)
                assert vec.vocabulary_ == vocab, kw
                assert_csr_identical(m_dev, sp.csr_matrix(m_ref), str(kw))
    # master + duplicates where only ONE of the two columns has non-ASCII characters: one alphabet for both
    master, dups = _names(200, seed=6), UNICODE_NAMES
    (a_ref, b_ref), vocab, _ = O.tfidf_sklearn(master + dups, [master, dups], dtype=dtype, normalize_to_ascii=False)
    vec = HipTfidfVectorizer(dtype=dtype, ctx=ctx, normalize_to_ascii=False)
    pm, pdu = vec.prepare(master), vec.prepare(dups)
    vec.fit_prepared([pm, pdu])
    assert vec.vocabulary_ == vocab
    assert_csr_identical(vec.transform_prepared(pm).to_scipy(), sp.csr_matrix(a_ref))
    assert_csr_identical(vec.transform_prepared(pdu).to_scipy(), sp.csr_matrix(b_ref))


def test_vectoriser_strings_of_any_length(ctx):
    """Strings with more n-grams than one wave sorts in LDS (1024) take the workgroup-per-string kernel; the reference
    has no length limit (a description or address column among the names must not abort the job)."""
    from string_grouper_amd.vectorizer import HipTfidfVectorizer
    rng = np.random.default_rng(3)
    words = _names(400, seed=9)

    def text(n_chars):
        out = []
        while sum(len(w) + 1 for w in out) < n_chars:
            out.append(words[int(rng.integers(len(words)))])
        return " ".join(out)[:n_chars]
    names = _names(500, seed=8) + [text(1023), text(1026), text(1500), text(5000), text(70000), "é" + text(2000) + "Ü", "x" * 3000,
                                    text(1025).lower()]
    for dtype in (np.float32, np.float64):
        for kw in (dict(), dict(ngram_size=5), dict(normalize_to_ascii=False)):
            (m_ref,), vocab, _ = O.tfidf_sklearn(names, [names], dtype=dtype, **kw)
            vec = HipTfidfVectorizer(dtype=dtype, ctx=ctx, **kw)
            m_dev = vec.fit_transform(names)
            assert vec.vocabulary_ == vocab, kw
            assert_csr_identical(m_dev, sp.csr_matrix(m_ref), str(kw))
            assert_csr_identical(vec.transform(names[-9:]), sp.csr_matrix(m_ref[-9:]), "transform of the long strings alone")


def test_vectoriser_wide_keys_take_the_sorted_vocabulary(ctx, monkeypatch):
    """n-gram keys wider than 30 bits (long n-grams, large alphabets) cannot index a dense table: 64-bit keys, the
    vocabulary is the sorted array of the distinct keys.  Also forced on ordinary 3-grams (SG_VOCAB_SORTED=1)."""
    from string_grouper_amd.vectorizer import HipTfidfVectorizer
    names = _names(3000, seed=13) + UNICODE_NAMES
    cases = [dict(ngram_size=6), dict(ngram_size=8), dict(ngram_size=10), dict(ngram_size=3, normalize_to_ascii=False),
             dict(ngram_size=4, normalize_to_ascii=False, ignore_case=False), dict(ngram_size=5, normalize_to_ascii=False),
             dict(ngram_size=8, normalize_to_ascii=False, ignore_case=False)]
    for dtype in (np.float32, np.float64):
        for kw in cases:
            (m_ref,), vocab, _ = O.tfidf_sklearn(names, [names], dtype=dtype, **kw)
            vec = HipTfidfVectorizer(dtype=dtype, ctx=ctx, **kw)
            m_dev = vec.fit_transform(names)
            assert vec.vocabulary_ == vocab, kw
            assert_csr_identical(m_dev, sp.csr_matrix(m_ref), str(kw))
    monkeypatch.setenv("SG_VOCAB_SORTED", "1")
    (m_ref,), vocab, _ = O.tfidf_sklearn(names, [names], dtype=np.float32)
    vec = HipTfidfVectorizer(dtype=np.float32, ctx=ctx)
    assert ctx.vocab_coding(vec.fit(names)._vocab)[2] is True
    assert vec.vocabulary_ == vocab
    assert_csr_identical(vec.transform(names), sp.csr_matrix(m_ref))


def test_public_api_on_code_point_ngrams(ctx):
    """match_strings / group_similar_strings with normalize_to_ascii=False (string_grouper.py:202) on names with
    non-ASCII characters: the HIP engine against the host mirror on the oracle engine."""
    import pandas as pd
    import string_grouper_amd as sga
    import string_grouper_amd.engine as E
    from tests._oracle_engine import OracleEngine
    names = pd.Series(UNICODE_NAMES * 3 + _names(2000, seed=21) + ["東京ホールディングス", "東京ホールディングス株式", "Crème Brûlée Holding"], name="name")
    old = E._engine
    try:
        out = {}
        for label, eng in (("hip", E.HipEngine(ctx)), ("oracle", OracleEngine(use_port=True))):
            E.set_engine(eng)
            out[label] = [sga.match_strings(names, normalize_to_ascii=False, min_similarity=0.6),
                          sga.match_strings(names, normalize_to_ascii=False, ignore_case=False, min_similarity=0.6,
                                            tfidf_matrix_dtype=np.float32),
                          sga.group_similar_strings(names, normalize_to_ascii=False, min_similarity=0.6)]
        for a, b in zip(out["hip"], out["oracle"]):
            pd.testing.assert_frame_equal(pd.DataFrame(a), pd.DataFrame(b))
    finally:
        E.set_engine(old)


def test_transform_of_strings_with_characters_unseen_at_fit(ctx):
    """sklearn's transform() drops the n-grams it has not seen at fit(); with ngram_size >= 4 the device codes
    characters by their rank among the bytes seen at fit(), so an unseen character needs its own code."""
    from sklearn.feature_extraction.text import TfidfVectorizer
    from string_grouper_amd.vectorizer import HipTfidfVectorizer
    fit_on = ["abcdefgh ijkl", "abcd efgh", "hgfedcba lkji", "abcabcabc"] * 5
    later = ["abcdefgh", "abcdxefgh", "zzzzzzzz", "abcd1234efgh", "ABCDEFGH-QRSTUVWXYZ", "", "a", "abcdefg hijklm nopq"]
    for n in (3, 4, 5, 6):
        for dtype in (np.float32, np.float64):
            ref = TfidfVectorizer(min_df=1, analyzer=lambda s, n=n: O.ngrams(s, n), dtype=dtype).fit(fit_on)
            vec = HipTfidfVectorizer(ngram_size=n, dtype=dtype, ctx=ctx).fit(fit_on)
            assert vec.vocabulary_ == ref.vocabulary_
            assert_csr_identical(vec.transform(later), sp.csr_matrix(ref.transform(later)), f"n={n}")
            assert_csr_identical(vec.transform(fit_on), sp.csr_matrix(ref.transform(fit_on)), f"n={n} fit set")


def test_run_twice_is_bitwise_identical(ctx, mats):
    """Determinism: per-row independence + fixed summation order (no race shows up as a diff)."""
    from string_grouper_amd.sparse_dot_topn import sp_matmul_topn
    A = mats[np.float32]
    C1 = sp_matmul_topn(A, A.T, 10, 0.7, sort=True, ctx=ctx)
    C2 = sp_matmul_topn(A, A.T, 10, 0.7, sort=True, ctx=ctx)
    assert_csr_identical(C1, C2)


def test_100k_selfjoin_matches_port(ctx):
    """BASELINE.json configs[1]: 100k synthetic names, 3-gram TF-IDF, ntop=10, min_sim=0.8, fp32."""
    from string_grouper_amd.sparse_dot_topn import sp_matmul_topn
    A = _tfidf(_names(100000), np.float32)
    C_dev = sp_matmul_topn(A, A.T, 10, 0.8, sort=True, ctx=ctx)
    C_ref = P.sp_matmul_topn_port(A, A.T, 10, 0.8, True, 16)
    assert_csr_identical(C_dev, C_ref)
    # size-independent properties: diagonal present with score ~1
    d = C_dev.diagonal()
    assert (np.abs(d[d > 0] - 1) < 1e-5).all()


def test_headline_663k_selfjoin_every_row_equals_sklearn_and_the_port(ctx):
    """BASELINE.json's headline configuration (configs[2] on the SynthNames-v1 stand-in): 663 000 names,
    3-grams, ntop=10, min_sim=0.8, fp32.  The device TF-IDF matrix against sklearn driven as the reference
    drives it, and ALL rows of the multiply (pruned kernel, symmetric mode -- the path bench.py times) against
    the C restatement of sp_matmul_topn; then the one-sided pruned kernel and the size-independent properties."""
    import os
    from string_grouper_amd.vectorizer import HipTfidfVectorizer
    n = 663000
    names = _names(n)
    vec = HipTfidfVectorizer(dtype=np.float32, ctx=ctx)
    prepared = vec.prepare(names)
    vec.fit_prepared([prepared])
    dA = vec.transform_prepared(prepared)
    A_dev = dA.to_scipy()
    A_ref = _tfidf(names, np.float32)
    assert_csr_identical(A_dev, A_ref, "tf-idf at 663k")
    threads = max(1, min(64, len(os.sched_getaffinity(0))))
    C_ref = P.sp_matmul_topn_port(A_ref, A_ref.T, 10, 0.8, True, threads)
    post = ctx.postings_build(dA)
    res = ctx.spgemm_topn(dA, post, 10, 0.8, True)
    st = ctx.stats()
    C_sym = res.to_scipy()
    res.free()
    assert st["prune_rows"] > 0 and st["exact_rows"] == 0
    assert_csr_identical(C_sym, C_ref, "663k self-join, symmetric pruned multiply")
    assert st["prune_postings"] < 0.06 * st["macs"]       # the symmetric pass streams half of the one-sided 8 %
    ctx.set_option("SG_SYM", "0")
    try:
        res = ctx.spgemm_topn(dA, post, 10, 0.8, True)
        C_one = res.to_scipy()
        res.free()
    finally:
        ctx.set_option("SG_SYM", None)
    assert_csr_identical(C_one, C_ref, "663k self-join, one-sided pruned multiply")
    post.free()
    # properties that hold at any size: every non-empty row finds itself with a score of 1 (+- rounding), rows are
    # ordered by (score desc, column asc), scores are in (0.8, 1 + eps]
    nnz_row = np.diff(A_ref.indptr) > 0
    d = C_sym.diagonal()
    full = np.diff(C_sym.indptr) == 10         # a hub of identical names: the ten lowest columns win the tie, maybe not i
    assert (d[~nnz_row] == 0).all() and ((np.abs(d - 1) < 1e-5) | full | ~nnz_row).all()
    assert (C_sym.data > np.float32(0.8)).all() and (C_sym.data < 1.0001).all()
    rows = np.repeat(np.arange(n), np.diff(C_sym.indptr))
    same_row = rows[1:] == rows[:-1]
    ok = (C_sym.data[:-1] > C_sym.data[1:]) | ((C_sym.data[:-1] == C_sym.data[1:]) & (C_sym.indices[:-1] < C_sym.indices[1:]))
    assert ok[same_row].all()


def test_results_do_not_depend_on_uninitialised_memory(ctx, monkeypatch):
    """With SG_POISON_ALLOC=1 the library fills every block it hands out with 0xFF bytes (NaN / -1) first.  Round 2's
    663k test found a score that was multiplied with the uninitialised pad behind the last packed row (0 * NaN);
    this runs the whole path -- odd sizes, self-join and two series, both dtypes, both forms of the pruned kernel,
    the exact kernel, the fused tail and the reductions -- on poisoned memory against the oracle."""
    import pandas as pd
    import string_grouper_amd as sga
    import string_grouper_amd.engine as E
    from string_grouper_amd.vectorizer import HipTfidfVectorizer
    monkeypatch.setenv("SG_POISON_ALLOC", "1")
    ctx.trim()                                   # nothing cached from earlier tests
    for n, dtype in ((4097, np.float32), (5001, np.float64), (12289, np.float32)):
        names = _names(n, seed=n)
        A_ref = _tfidf(names, dtype)
        C_ref = P.sp_matmul_topn_port(A_ref, A_ref.T, 10, 0.8, True, 8)
        for env in ({}, {"SG_SYM": "1"}, {"SG_PRUNE": "0"}):
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            vec = HipTfidfVectorizer(dtype=dtype, ctx=ctx)
            p = vec.prepare(names)
            vec.fit_prepared([p])
            dA = vec.transform_prepared(p)
            assert_csr_identical(dA.to_scipy(), A_ref, f"tf-idf n={n}")
            post = ctx.postings_build(dA)
            res = ctx.spgemm_topn(dA, post, 10, 0.8, True)
            assert_csr_identical(res.to_scipy(), C_ref, f"n={n} {dtype.__name__} {env}")
            res.free()
            post.free()
            dA.free()
            for k in env:
                monkeypatch.delenv(k)
    old = E._engine
    E.set_engine(E.HipEngine(ctx))
    try:
        from tests import _golden as G
        G.run_api_checks(sga)
        s = pd.Series(_names(3001, seed=3))
        a = sga.group_similar_strings(s, min_similarity=0.8)
        b = sga.match_most_similar(s, pd.Series(_names(1001, seed=4)), min_similarity=0.7)
        monkeypatch.delenv("SG_POISON_ALLOC")
        pd.testing.assert_frame_equal(pd.DataFrame(a), pd.DataFrame(sga.group_similar_strings(s, min_similarity=0.8)))
        pd.testing.assert_frame_equal(pd.DataFrame(b), pd.DataFrame(sga.match_most_similar(s, pd.Series(_names(1001, seed=4)),
                                                                                        min_similarity=0.7)))
    finally:
        E.set_engine(old)


def test_public_api_golden_cases_on_gpu(ctx):
    """The reference's golden vectors (tests/golden, made by the unmodified reference) through the
    drop-in API with the HIP engine: match frames, groups, most-similar, known answers."""
    import string_grouper_amd as sga
    import string_grouper_amd.engine as E
    from tests import _golden as G
    old = E._engine
    E.set_engine(E.HipEngine(ctx))
    try:
        G.run_api_checks(sga)
    finally:
        E.set_engine(old)


def test_blocked_equals_unblocked_on_gpu(ctx, mats):
    """n_blocks only cuts the work differently (reference invariant, test_string_grouper.py:191-336)."""
    import string_grouper_amd.engine as E
    eng = E.HipEngine(ctx)
    A = eng.wrap(mats[np.float64][:3000])
    B = eng.wrap(mats[np.float64][:5000])
    ref = eng.topn_multiply(A, B, 8, 0.4).astype(np.float64)
    for nb in ((1, 3), (2, 1), (3, 4)):
        got = eng.topn_multiply_blocked(A, B, nb, 8, 0.4)
        assert_csr_identical(got, ref, str(nb))


def test_device_resident_inputs_and_rccl_plumbing(ctx):
    """sg_strings_from_device / sg_csr_from_device / zero-copy torch views, and the single-rank form
    of the multi-GPU path (RCCL process group of size 1 on this GPU)."""
    import os
    import torch
    import torch.distributed as dist
    from string_grouper_amd import distributed as D
    from string_grouper_amd.vectorizer import HipTfidfVectorizer
    names = _names(6000, seed=11)
    vec = HipTfidfVectorizer(dtype=np.float32, ctx=ctx)
    p = vec.prepare(names)
    # strings handed over as device pointers (torch tensors own the memory)
    t_bytes = torch.from_numpy(p.data.copy()).cuda()
    t_offs = torch.from_numpy(p.offsets.copy()).cuda()
    torch.cuda.synchronize()
    p2 = type(p)("bytes", p.data, p.offsets)
    p2.dev = ctx.strings_from_device(t_bytes.data_ptr(), t_offs.data_ptr(), p.n, int(p.offsets[-1]), keepalive=(t_bytes, t_offs))
    vec.fit_prepared([p2])
    A = vec.transform_prepared(p2)
    ctx.sync()
    (A_ref,), _, _ = O.tfidf_sklearn(names, [names], dtype=np.float32)
    assert_csr_identical(A.to_scipy(), sp.csr_matrix(A_ref))
    # zero-copy torch view of the library's CSR and back
    ip, ix, d = D.csr_as_torch(A)
    assert ip.shape[0] == len(names) + 1 and int(ip[-1]) == A_ref.nnz
    A2 = D.csr_from_torch(ctx, ip.clone(), ix.clone(), d.clone(), A.dims()[:2])
    torch.cuda.synchronize()
    assert_csr_identical(A2.to_scipy(), sp.csr_matrix(A_ref))
    # the sharded driver with a one-rank RCCL group
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        res, (lo, hi), n_total = D.sharded_self_join(ctx, p, lambda: HipTfidfVectorizer(dtype=np.float32, ctx=ctx), 10, 0.8)
        assert (lo, hi, n_total) == (0, len(names), len(names))
        C_ref = P.sp_matmul_topn_port(A_ref, A_ref.T, 10, 0.8, True, 4)
        assert_csr_identical(res.to_scipy(), C_ref)
        counts = D.gather_counts(torch.from_numpy(np.diff(C_ref.indptr).astype(np.int32)).cuda(), n_total)
        assert counts.cpu().numpy().tolist() == np.diff(C_ref.indptr).tolist()
        # string-broadcast form (what bench.py uses for N > 1)
        tb, to = D.strings_to_device_tensors(p, torch.device("cuda", 0))
        local, tb2, to2 = D.broadcast_strings(ctx, tb, to)
        res2, blk, n2 = D.sharded_self_join_replicated(ctx, local, lambda: HipTfidfVectorizer(dtype=np.float32, ctx=ctx), 10, 0.8)
        assert blk == (0, len(names)) and n2 == len(names)
        assert_csr_identical(res2.to_scipy(), C_ref)
        # round 2's sharded form on the one rank: local block = everything; the df table, the CSR and the result are
        # handed to torch as zero-copy views, pushed through the (1-rank) RCCL collectives and wrapped again
        ops = D.HipOps(ctx, lambda: HipTfidfVectorizer(dtype=np.float32, ctx=ctx))
        block, (lo, hi) = D.local_string_block(ctx, tb2, to2, 0, 1)
        assert (lo, hi) == (0, len(names))
        res3, vec3 = D.distributed_self_join(ops, block, 10, 0.8)
        assert_csr_identical(res3.to_scipy(), C_ref)
        np.testing.assert_array_equal(vec3.idf_, O.tfidf_sklearn(names, [names], dtype=np.float32)[2])
        cols, vals, counts = D.gather_topn(ops, res3)
        assert counts.tolist() == np.diff(C_ref.indptr).tolist()
        # the self-join form over row ranges, forced on the one rank: range = everything, the pair list goes through the
        # (1-rank) all-gather and comes back for the merge
        os.environ["SG_DIST_SYM"] = "1"
        try:
            res5, _ = D.distributed_self_join(ops, block, 10, 0.8)
            assert isinstance(res5, D.TopNRows) and ctx.stats()["prune_symmetric"] == 1
            C5 = res5.to_scipy()              # the rank's block: rows in POSITION order when the index is permuted
            if res5.orig_of is not None:
                C5 = C5[np.argsort(res5.orig_of.cpu().numpy(), kind="stable")]
            assert_csr_identical(C5, C_ref)
            c5, v5, n5 = D.gather_topn(ops, res5)
            assert n5.tolist() == np.diff(C_ref.indptr).tolist()
        finally:
            os.environ["SG_DIST_SYM"] = "0"
        ipg, ixg, dg, shp = D.all_gather_csr(*D.csr_as_torch(A), A.dims()[1])     # the collective form, explicitly
        A3 = ops.csr_from_tensors(ipg, ixg, dg, shp)
        assert_csr_identical(A3.to_scipy(), sp.csr_matrix(A_ref))
        # master x duplicates through the same path
        dups = _names(2500, seed=12)
        pd_ = vec.prepare(dups)
        res4, _ = D.distributed_match(ops, p, pd_, 20, 0.7)
        (Am, Bd), _, _ = O.tfidf_sklearn(names + dups, [names, dups], dtype=np.float32)
        assert_csr_identical(res4.to_scipy(), P.sp_matmul_topn_port(Am, Bd.T, 20, 0.7, True, 4))
        # and behind the public API: the distributed engine on a 1-rank group must give what the plain engine gives
        import pandas as pd
        import string_grouper_amd as sga
        import string_grouper_amd.engine as E
        old = E._engine
        try:
            s_m, s_d = pd.Series(names[:3000], name="name"), pd.Series(dups[:1000], name="dup")
            E.set_engine(E.HipEngine(ctx))
            want = [sga.match_strings(s_m, min_similarity=0.8, max_n_matches=10, tfidf_matrix_dtype=np.float32),
                    sga.match_strings(s_m, s_d, min_similarity=0.7), sga.group_similar_strings(s_m, min_similarity=0.8),
                    sga.match_most_similar(s_m, s_d, min_similarity=0.7)]
            E.set_engine(E.DistributedHipEngine(ctx))
            got = [sga.match_strings(s_m, min_similarity=0.8, max_n_matches=10, tfidf_matrix_dtype=np.float32),
                   sga.match_strings(s_m, s_d, min_similarity=0.7), sga.group_similar_strings(s_m, min_similarity=0.8),
                   sga.match_most_similar(s_m, s_d, min_similarity=0.7)]
            for w, g in zip(want, got):
                pd.testing.assert_frame_equal(pd.DataFrame(w), pd.DataFrame(g))
            # a list long enough, and repetitive enough, for the library to index identical names once: the distributed
            # engine's self-join then runs over ranges of GROUPS, the rank expands its groups into rows, and the gathered
            # rows are put in place by their numbers (distributed.gather_topn) -- forced onto the one rank
            big = list(_names(12000, seed=13))
            s_b = pd.Series(big + big[:700] + [big[5]] * 300, name="name")
            E.set_engine(E.HipEngine(ctx))
            want_b = sga.match_strings(s_b, min_similarity=0.8, max_n_matches=10, tfidf_matrix_dtype=np.float32)
            os.environ["SG_DIST_SYM"] = "1"
            try:
                E.set_engine(E.DistributedHipEngine(ctx))
                got_b = sga.match_strings(s_b, min_similarity=0.8, max_n_matches=10, tfidf_matrix_dtype=np.float32)
            finally:
                os.environ["SG_DIST_SYM"] = "0"
            pd.testing.assert_frame_equal(want_b, got_b)
            vb = HipTfidfVectorizer(dtype=np.float32, ctx=ctx)
            pb = vb.prepare(list(s_b))
            vb.fit_prepared([pb])
            post_b = ctx.postings_build(vb.transform_prepared(pb))
            assert ctx.postings_rows(post_b)[0] < len(s_b) - 900, "the list was meant to be indexed over groups"
            post_b.free()
        finally:
            E.set_engine(old)
    finally:
        dist.destroy_process_group()


def _check_slice_properties(res, lo, n_left_block, n_right, top_n, thr, self_join, A_host, B_host, sample, what):
    """Size-independent properties of one row block of C plus an exact comparison of ``sample`` rows
    against the CPU port."""
    cols, vals, cnt = res.to_host()
    assert cols.shape == (n_left_block, top_n)
    assert cnt.min() >= 0 and cnt.max() <= top_n, what
    mask = np.arange(top_n)[None, :] < cnt[:, None]
    assert (vals[mask] > np.float32(thr)).all(), what                      # strictly above the threshold
    assert ((cols[mask] >= 0) & (cols[mask] < n_right)).all(), what
    v = np.where(mask, vals, -np.inf)
    assert (v[:, :-1] >= v[:, 1:]).all(), what                             # score descending within a row
    ties = mask[:, 1:] & (v[:, :-1] == v[:, 1:])
    assert (cols[:, :-1][ties] < cols[:, 1:][ties]).all(), what            # ties: column ascending
    if self_join:                                                          # every non-empty row matches itself ~1
        rows_with_grams = np.diff(A_host.indptr)[lo:lo + n_left_block] > 0
        own = (cols == (np.arange(n_left_block) + lo)[:, None]) & mask
        full = cnt == top_n
        assert (own.any(axis=1) | full | ~rows_with_grams).all(), what
        assert np.abs(vals[own] - 1.0).max() < 1e-5, what
    rng = np.random.default_rng(5)
    pick = np.sort(rng.choice(n_left_block, size=min(sample, n_left_block), replace=False))
    C_ref = P.sp_matmul_topn_port(A_host[lo + pick], B_host.T, top_n, thr, True, 32)
    for r, i in enumerate(pick):
        ref_c = C_ref.indices[C_ref.indptr[r]:C_ref.indptr[r + 1]]
        ref_v = C_ref.data[C_ref.indptr[r]:C_ref.indptr[r + 1]]
        assert cnt[i] == len(ref_c), f"{what}: row {lo + i} count"
        assert np.array_equal(cols[i, :cnt[i]], ref_c) and np.array_equal(vals[i, :cnt[i]], ref_v), f"{what}: row {lo + i}"


def _all_rows_equal_the_exact_kernel(ctx, res, A, B, top_n, thr, what):
    """Full size, ALL rows: the product path's result against the exact kernel K4 run on every row of the same matrices
    (no pruning, identical rows not grouped) -- the kernel that the sampled rows and the smaller sizes pin on the CPU port
    (the port itself on every row of these sizes would take a quarter of an hour of the box's host cores)."""
    ctx.set_option("SG_PRUNE", "0")
    ctx.set_option("SG_COLLAPSE", "0")
    post = ctx.postings_build(B)
    ex = ctx.spgemm_topn(A, post, top_n, thr, True)
    st = ctx.stats()
    ctx.reset_options()
    assert st["prune_rows"] == 0, what
    c0, v0, n0 = res.to_host()
    c1, v1, n1 = ex.to_host()
    ex.free()
    post.free()
    np.testing.assert_array_equal(n0, n1, err_msg=what)
    mask = np.arange(c0.shape[1])[None, :] < n0[:, None]
    assert np.array_equal(c0[mask], c1[mask]) and np.array_equal(v0[mask], v1[mask]), what
    print(f"{what}: all {len(n0)} rows, {int(n0.sum())} matches, identical to the exact kernel ({st['ms_spgemm_topn']:.0f} ms)")


def _rows_equal_the_port(res, rows, A_host, B_host, top_n, thr, what):
    """The rows ``rows`` (ascending row numbers) of a full-size result against the CPU port on all host cores, bit for bit."""
    import os
    import time
    cols, vals, cnt = res.to_host()
    rows = np.unique(np.asarray(rows, dtype=np.int64))
    n_cpu = min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 8), 64)
    t0 = time.perf_counter()
    # the reference's own scheme for a right-hand side of this size (string_grouper.py:733-746): column blocks of B, each
    # multiplied on its own, the blocks' top-n lists zipped -- a per-thread accumulator over ALL 5 M columns (40 MB of
    # randomly touched memory per thread) made the port 100 x slower per product than at 663 k
    from oracle.ref_pipeline import zip_port
    left = A_host[rows]
    step = 262_144
    parts = [P.sp_matmul_topn_port(left, B_host[b0:b0 + step].T, top_n, thr, True, n_cpu) for b0 in range(0, B_host.shape[0], step)]
    C_ref = zip_port(top_n, parts)
    secs = time.perf_counter() - t0
    got_cnt = cnt[rows]
    np.testing.assert_array_equal(got_cnt, np.diff(C_ref.indptr), err_msg=f"{what}: match counts")
    mask = np.arange(cols.shape[1])[None, :] < got_cnt[:, None]
    assert np.array_equal(cols[rows][mask], C_ref.indices), f"{what}: columns"
    assert np.array_equal(vals[rows][mask], C_ref.data), f"{what}: scores"
    print(f"{what}: {len(rows)} rows ({int(got_cnt.sum())} matches) identical to the CPU port ({secs:.1f} s on {n_cpu} cores)")
    return len(rows)


@pytest.mark.timeout(900)
def test_forms_outside_the_name_matching_envelope_at_300k_equal_the_exact_kernel_and_the_port(ctx):
    """Round 6 at a size where the forms run as they are CHOSEN (300 000 names: self-join form, identical rows grouped, the
    second index built on first use): top 100 at 0.8 (two register lists in the second pass), top 20 at 0.5 and top 10 at 0.62
    (tile-by-tile form), top 10 at 0.35 (exact kernel in the self-join form on its own layout) -- every row against the
    one-sided exact kernel (SG_PRUNE=0, SG_EXACT_SYM=0), 20 000 rows of each against the CPU port."""
    from string_grouper_amd.vectorizer import HipTfidfVectorizer
    names = _names(300000, seed=17)
    vec = HipTfidfVectorizer(dtype=np.float32, ctx=ctx)
    p = vec.prepare(names)
    vec.fit_prepared([p])
    dA = vec.transform_prepared(p)
    A_host = dA.to_scipy()
    rows = np.concatenate([np.arange(7000), 150000 + np.arange(7000), 293000 + np.arange(7000)])
    post = ctx.postings_build(dA)
    for top_n, thr, pruned in ((100, 0.8, True), (20, 0.5, True), (10, 0.62, True), (10, 0.35, False)):
        res = ctx.spgemm_topn(dA, post, top_n, thr, True)
        st = ctx.stats()
        what = f"300 k names, top {top_n} at {thr}"
        assert st["prune_symmetric"] == 1 and (st["prune_rows"] > 0) == pruned and (st["exact_rows"] == 0) == pruned, (what, st)
        got = res.to_host()
        ctx.set_option("SG_PRUNE", "0")
        ctx.set_option("SG_EXACT_SYM", "0")
        ref = ctx.spgemm_topn(dA, post, top_n, thr, True)
        assert ctx.stats()["prune_symmetric"] == 0 and ctx.stats()["prune_rows"] == 0
        want = ref.to_host()
        ctx.reset_options()
        assert np.array_equal(got[2], want[2]), what + ": match counts"
        mask = np.arange(want[0].shape[1])[None, :] < want[2][:, None]
        assert np.array_equal(got[0][mask], want[0][mask]) and np.array_equal(got[1][mask], want[1][mask]), what + ": every row vs the exact kernel"
        _rows_equal_the_port(res, rows, A_host, A_host, top_n, thr, what)
        res.free()
        ref.free()
    post.free()
    dA.free()


def test_frames_of_the_public_api_with_and_without_the_host_helpers_at_300k(ctx):
    """Round 6: `match_strings` at a size where the host helpers take over (libsg_host.so: the object gathers, the match
    list's rows expanded, columns and scores widened, the index labels) against the same call on numpy / pandas alone -- equal
    frames, column by column and dtype by dtype; with a non-default index and ids as well."""
    import pandas as pd
    import string_grouper_amd as sga
    import string_grouper_amd.engine as E
    from string_grouper_amd import _hostops as H
    if H._load() is None:
        pytest.skip("libsg_host.so not built")
    E.set_engine(E.HipEngine(ctx))
    names = pd.Series(_names(300000, seed=23))
    ids = pd.Series(np.arange(len(names)) * 3 + 7)
    shifted = names.copy()
    shifted.index = pd.RangeIndex(10, 10 + 2 * len(names), 2)
    for dt in (np.float32, np.float64):
        frames = {}
        for helpers in (True, False):
            saved = H._lib
            if not helpers:
                H._lib = None
            try:
                frames[helpers] = (sga.match_strings(names, max_n_matches=10, min_similarity=0.8, tfidf_matrix_dtype=dt),
                                   sga.match_strings(shifted, master_id=ids, max_n_matches=10, min_similarity=0.8, tfidf_matrix_dtype=dt))
            finally:
                H._lib = saved
        for a, b in zip(frames[True], frames[False]):
            assert len(a) > 500000 and list(a.columns) == list(b.columns) and (a.dtypes == b.dtypes).all()
            assert a.equals(b), dt


def _device_u32(ptr, n):
    """Host copy of a uint32 device array of the library (test plumbing: a torch view of the pointer)."""
    import torch
    from string_grouper_amd import distributed as D
    return torch.as_tensor(D.DeviceTensorView(ptr, n, "<u4"), device=torch.device("cuda", 0)).cpu().numpy().astype(np.int64)


def _rows_at_position_blocks(ctx, post, n_rows, width):
    """Rows whose position in the index (of their group of identical rows, if grouped) lies in the first, the middle or the
    last ``width`` positions: the self-join form walks positions from the last down, and a position's cost grows with it."""
    n_index, n_caller, p_gid = ctx.postings_rows(post)
    assert n_caller == n_rows
    p_orig, p_pos = ctx.postings_permutation(post)
    ctx.sync()
    gid = _device_u32(p_gid, n_rows) if p_gid else np.arange(n_rows)
    pos_of = _device_u32(p_pos, n_index) if p_pos else np.arange(n_index)
    row_pos = pos_of[gid]
    mid = n_index // 2
    take = (row_pos < width) | ((row_pos >= mid) & (row_pos < mid + width)) | (row_pos >= n_index - width)
    return np.flatnonzero(take)


def _rows_of_special_shape(A_host, n_short=3000):
    """Rows the pruned kernel treats apart: more than 64 non-zeros (the wide launch; beyond 128: the exact kernel) and the
    shortest rows (few, frequent terms, all of them in the prefix: the rows of thousands of rounds, set aside as parts)."""
    nnz = np.diff(A_host.indptr)
    wide = np.flatnonzero(nnz > 64)
    short = np.flatnonzero((nnz > 0) & (nnz <= 4))[:n_short]
    return np.concatenate([wide, short])


@pytest.mark.timeout(900)
def test_config4_5M_selfjoin_one_of_eight_row_blocks(ctx):
    """BASELINE.json configs[3]: 5M synthetic names self-join, left CSR row-blocked over 8 GPUs -- here
    the block rank 3 would own, against the full right-hand side (postings exceed the Infinity Cache,
    so the tile-group path runs).  Properties at full size + sampled rows against the CPU port."""
    from string_grouper_amd import distributed as D
    from string_grouper_amd.vectorizer import HipTfidfVectorizer
    n = 5_000_000
    names = _names(n, seed=1234)
    vec = HipTfidfVectorizer(dtype=np.float32, ctx=ctx)
    p = vec.prepare(names)
    vec.fit_prepared([p])
    A = vec.transform_prepared(p)
    A_host = A.to_scipy()
    post = ctx.postings_build(A)
    lo, hi = D.row_block(3, 8, n)
    blk = A.row_block(lo, hi)
    res = ctx.spgemm_topn(blk, post, 10, 0.8, True)
    st = ctx.stats()
    print(f"config4 block: rows {hi - lo}, K4 {st['ms_spgemm_topn']:.1f} ms, macs {st['macs']:.3e}, "
          f"{st['spgemm_bytes'] / st['ms_spgemm_topn'] / 1e9:.2f} TB/s (stream model of the exact kernel); pruned rows "
          f"{st['prune_rows']}, postings streamed {st['prune_postings']:.3e}, pairs scored {st['prune_survivors']:.3e}")
    _check_slice_properties(res, lo, hi - lo, n, 10, 0.8, True, A_host, A_host, 300, "config4")
    res.free()
    blk.free()
    # the whole self-join on this one GPU (self-join form on the groups of identical rows):
    res = ctx.spgemm_topn(A, post, 10, 0.8, True)
    assert ctx.stats()["prune_symmetric"] == 1
    # (1) more than 100 000 rows against the CPU port -- the first, middle and last positions of the index and the rows of
    #     special shape -- on all host cores (300 sampled rows were inference, not evidence)
    rows = np.concatenate([_rows_at_position_blocks(ctx, post, n, 30_000), _rows_of_special_shape(A_host)])
    assert _rows_equal_the_port(res, rows, A_host, A_host, 10, 0.8, "configs[3], 5 M self-join") >= 100_000
    del A_host
    # (2) every row against the exact kernel
    _all_rows_equal_the_exact_kernel(ctx, res, A, A, 10, 0.8, "configs[3], 5 M self-join")
    for h in (res, post, A):
        h.free()
    ctx.trim()


@pytest.mark.timeout(900)
def test_config5_asymmetric_10M_x_1M_one_of_eight_row_blocks(ctx):
    """BASELINE.json configs[4]: master 10M x duplicates 1M, ntop=20, min_sim=0.7 (match_most_similar
    path: top-n per MASTER row, string_grouper.py:728-729) -- one of the eight master row blocks."""
    from string_grouper_amd import distributed as D
    from string_grouper_amd.synth import synth_names
    from string_grouper_amd.vectorizer import HipTfidfVectorizer
    n_m, n_d = 10_000_000, 1_000_000
    master = _names(n_m, seed=1234)
    dupes = synth_names(n_d, seed=4321, perturb_of=master, perturb_frac=0.5)
    vec = HipTfidfVectorizer(dtype=np.float32, ctx=ctx)
    pm, pd_ = vec.prepare(master), vec.prepare(dupes)
    del master
    vec.fit_prepared([pm, pd_])
    A = vec.transform_prepared(pm)
    B = vec.transform_prepared(pd_)
    B_host = B.to_scipy()
    post = ctx.postings_build(B)
    lo, hi = D.row_block(5, 8, n_m)
    blk = A.row_block(lo, hi)
    res = ctx.spgemm_topn(blk, post, 20, 0.7, True)
    st = ctx.stats()
    print(f"config5 block: rows {hi - lo}, K4 {st['ms_spgemm_topn']:.1f} ms, macs {st['macs']:.3e}; pruned rows "
          f"{st['prune_rows']}, postings streamed {st['prune_postings']:.3e}, pairs scored {st['prune_survivors']:.3e}")
    A_blk_host = blk.to_scipy()
    # _check_slice_properties indexes the left matrix with lo + pick: hand it a matrix whose row 0 is row lo
    class _Shift:
        def __init__(self, m, lo):
            self.m, self.lo, self.indptr = m, lo, None
        def __getitem__(self, idx):
            return self.m[np.asarray(idx) - self.lo]
    _check_slice_properties(res, lo, hi - lo, n_d, 20, 0.7, False, _Shift(A_blk_host, lo), B_host, 300, "config5")
    res.free()
    blk.free()
    del A_blk_host
    # all 10 M master rows on this one GPU (identical master rows are multiplied once: round 4):
    res = ctx.spgemm_topn(A, post, 20, 0.7, True)
    st = ctx.stats()
    print(f"configs[4] whole: K4p group {st['ms_spgemm_topn']:.1f} ms, left rows multiplied {st['prune_rows']} of {n_m}")
    # (1) more than 100 000 master rows against the CPU port: the first, middle and last 34 000 and the rows of special shape
    A_host = A.to_scipy()
    rows = np.concatenate([np.arange(34_000), n_m // 2 + np.arange(34_000), n_m - 34_000 + np.arange(34_000),
                           _rows_of_special_shape(A_host)])
    assert _rows_equal_the_port(res, rows, A_host, B_host, 20, 0.7, "configs[4], 10 M x 1 M") >= 100_000
    del A_host, B_host
    # (2) every row against the exact kernel
    _all_rows_equal_the_exact_kernel(ctx, res, A, B, 20, 0.7, "configs[4], 10 M x 1 M")
    for h in (res, post, A, B):
        h.free()
    ctx.trim()


def test_seeded_fuzz_of_the_public_api_hip_engine_equals_oracle_engine(ctx):
    """The 150 random small jobs of tests/_fuzz_cases.py (options drawn at random; all five entry points; on the CPU they
    are compared with the unmodified reference through the oracle engine, tests/test_host_api.py) through the HIP
    engine: frames, Series and exceptions must be what the oracle engine gives, bit for bit."""
    import pandas as pd
    import string_grouper_amd as sga
    import string_grouper_amd.engine as E
    from tests import _fuzz_cases as fz
    from tests._oracle_engine import OracleEngine
    old = E._engine
    try:
        for c, kind, m, dd, mid, did, kw in fz.cases():
            E.set_engine(OracleEngine())
            want = fz.run(sga, kind, m, dd, mid, did, kw)
            E.set_engine(E.HipEngine(ctx))
            got = fz.run(sga, kind, m, dd, mid, did, kw)
            what = f"case {c}: {kind} {kw}"
            if isinstance(want, tuple):
                assert isinstance(got, tuple) and got[:2] == want[:2], (what, got, want)
            elif isinstance(want, pd.DataFrame):
                pd.testing.assert_frame_equal(got, want, obj=what)
            else:
                pd.testing.assert_series_equal(got, want, obj=what)
    finally:
        E.set_engine(old)


def test_vectoriser_fuzz_against_python_semantics(ctx):
    """Seeded fuzz of K1/K2 against the reference's analyzer + sklearn: every ASCII byte (controls,
    punctuation, the \\x1c-\\x1f separators \\s matches), case, non-ASCII that NFKD folds to ASCII,
    folds to several characters, or drops entirely; empty strings and strings shorter than an n-gram."""
    from string_grouper_amd.vectorizer import HipTfidfVectorizer
    rng = np.random.default_rng(2024)
    ascii_pool = [chr(c) for c in range(1, 128)]
    uni_pool = list("ÀâÉèïÖüßÑçŁøÆœ") + ["ﬁ", "ﬂ", "Ⅻ", "½", "İ", "ı", "ǅ", "ẞ", "K", "Å", "中", "文", "😀", " ", " ", "́"]
    words = ["inc", "LLC", "Corp.", "holdings", "a.b.c", "X-Ray", "O'Neil", "  ", "\t", "GmbH & Co. KG"]
    strings = []
    for _ in range(4000):
        kind = rng.integers(0, 4)
        if kind == 0:
            n = int(rng.integers(0, 40))
            s = "".join(ascii_pool[int(x)] for x in rng.integers(0, len(ascii_pool), n))
        elif kind == 1:
            n = int(rng.integers(0, 30))
            s = "".join(uni_pool[int(rng.integers(0, len(uni_pool)))] if rng.random() < 0.3
                        else ascii_pool[int(rng.integers(31, 127))] for _ in range(n))
        elif kind == 2:
            s = " ".join(words[int(x)] for x in rng.integers(0, len(words), int(rng.integers(1, 6))))
        else:
            s = "".join(chr(int(x)) for x in rng.integers(97, 123, int(rng.integers(0, 130))))
        strings.append(s)
    strings += ["", "a", "ab", "abc", "ABC", "a b c", "...", "\x1c\x1d\x1e\x1f", "ǅ", "İi"]
    for kw in (dict(), dict(ignore_case=False), dict(ngram_size=2), dict(ngram_size=4)):
        for dtype in (np.float32, np.float64):
            (m_ref,), vocab, idf = O.tfidf_sklearn(strings, [strings], dtype=dtype, **kw)
            vec = HipTfidfVectorizer(dtype=dtype, ctx=ctx, **kw)
            m_dev = vec.fit(strings).transform(strings)
            assert vec.vocabulary_ == vocab, kw
            np.testing.assert_array_equal(vec.idf_, idf)
            assert_csr_identical(m_dev, sp.csr_matrix(m_ref), f"{kw} {dtype.__name__}")


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_fused_device_tail_equals_host_tail(ctx, dtype):
    """K6 (diagonal := 1, symmetrise, match list on the device) against the host implementation of the
    same reference semantics (string_grouper.py:417-431, :954-964, :755-763), self-join with hub rows
    (many exact duplicates, max_n_matches small so that mirrors overflow the per-row cap) and two-series."""
    import pandas as pd
    import string_grouper_amd as sga
    import string_grouper_amd.engine as E
    old = E._engine
    E.set_engine(E.HipEngine(ctx))
    try:
        names = _names(20000, seed=3) + ["ACME HOLDINGS INC"] * 300 + ["ACME HOLDING INC"] * 5 + ["", "ab"]
        s = pd.Series(names)
        for kw in (dict(max_n_matches=3, min_similarity=0.6), dict(max_n_matches=20, min_similarity=0.8)):
            sg_dev = sga.StringGrouper(s, tfidf_matrix_dtype=dtype, **kw)
            sg_dev.fit()
            sg_host = sga.StringGrouper(s, tfidf_matrix_dtype=dtype, **kw)
            sg_host._can_fuse_on_device = lambda: False
            sg_host.fit()
            pd.testing.assert_frame_equal(sg_dev._matches_list, sg_host._matches_list)
            assert sg_dev._true_max_n_matches == sg_host._true_max_n_matches
            # reference invariants: every string matches itself with similarity exactly 1, list is symmetric
            ml = sg_dev._matches_list
            diag = ml[ml.master_side == ml.dupe_side]
            assert len(diag) == len(s) and (diag.similarity == 1.0).all()
            fwd = set(zip(ml.master_side.tolist(), ml.dupe_side.tolist()))
            assert all((c, r) in fwd for r, c in fwd)
        d = pd.Series(_names(3000, seed=4))
        sg_dev = sga.StringGrouper(s, d, tfidf_matrix_dtype=dtype, max_n_matches=5, min_similarity=0.5).fit()
        sg_host = sga.StringGrouper(s, d, tfidf_matrix_dtype=dtype, max_n_matches=5, min_similarity=0.5)
        sg_host._can_fuse_on_device = lambda: False
        sg_host.fit()
        pd.testing.assert_frame_equal(sg_dev._matches_list, sg_host._matches_list)
    finally:
        E.set_engine(old)


def test_oversized_right_hand_side_is_split_and_zipped(ctx, mats, monkeypatch):
    """When one inverted index cannot hold the right-hand side the engine cuts it into blocks and merges
    on the device -- same result (forced here at a small size through the SG_MAX_POSTINGS test hook)."""
    import string_grouper_amd.engine as E
    eng = E.HipEngine(ctx)
    A = eng.wrap(mats[np.float32][:6000])
    B = eng.wrap(mats[np.float32])
    ref = eng.topn_multiply(A, B, 10, 0.6)
    monkeypatch.setenv("SG_MAX_POSTINGS", "120000")          # 20 k rows x 19 nnz = 380 k entries -> 4 blocks
    got = eng.topn_multiply(A, B, 10, 0.6)
    assert_csr_identical(got, ref)
    rows, cols, sims, tmax = eng.match_list(B, B, 10, 0.8, True)
    monkeypatch.delenv("SG_MAX_POSTINGS")
    rows2, cols2, sims2, tmax2 = eng.match_list(B, B, 10, 0.8, True)
    assert np.array_equal(rows, rows2) and np.array_equal(cols, cols2) and np.array_equal(sims, sims2) and tmax == tmax2


# ------------------------------------------------------------------------------------------------
# The pruned multiply (K4p, sg_spgemm_pruned.hip) must be indistinguishable from the exact kernel (K4)
# and from the oracle: same entries, same bits, same order.
def _multiply_both_ways(ctx, dA, dB, top_n, thr, monkeypatch, expect_sym=True, **env):
    """pruned ("1"), exact ("0") and -- self-joins -- the pruned kernel's self-join form forced at any size ("sym":
    by default it only runs from SG_SYM_MIN_ROWS rows on)."""
    out = {}
    for prune in ("1", "0") + (("sym",) if dA is dB else ()):
        monkeypatch.setenv("SG_PRUNE", "1" if prune == "sym" else prune)
        monkeypatch.setenv("SG_SYM", "1" if prune == "sym" else "0")
        for k, v in env.items():
            monkeypatch.setenv(k, str(v))
        post = ctx.postings_build(dB)
        res = ctx.spgemm_topn(dA, post, top_n, thr, True)
        st = ctx.stats()
        out[prune] = (res.to_scipy(), st)
        res.free()
        post.free()
    monkeypatch.delenv("SG_SYM")
    if "sym" in out:
        assert_csr_identical(out["sym"][0], out["0"][0], "self-join form vs exact")
        if expect_sym and out["1"][1]["prune_rows"] > 0 and out["1"][1]["exact_rows"] == 0:
            assert out["sym"][1]["prune_symmetric"] == 1
    return out


@pytest.mark.parametrize("family", ["4-grams", "master x duplicates", "low threshold", "short numeric strings"])
def test_pruned_multiply_keeps_its_lead_on_other_data_families(ctx, family, monkeypatch):
    """Performance guard outside the 663 k headline (kernel times from sg_stats; scripts/family_sweep.py has the full
    table): on these families the pruned multiply is 3-8 x faster than the exact kernel -- a regression of the filter
    (or a tuning that only fits SynthNames 3-grams) shows up here as a lost lead.  Results are compared as well."""
    from string_grouper_amd.vectorizer import HipTfidfVectorizer
    kw, top_n, thr = {}, 10, 0.8
    if family == "4-grams":
        master, dups, kw = _names(150000, seed=41), None, dict(ngram_size=4)
    elif family == "master x duplicates":
        master = _names(200000, seed=42)
        from string_grouper_amd.synth import synth_names
        dups, top_n, thr = synth_names(60000, 43, perturb_of=master, perturb_frac=0.5), 20, 0.7
    elif family == "low threshold":
        master, dups, top_n, thr = _names(150000, seed=44), None, 20, 0.6
    else:
        rng = np.random.default_rng(45)
        master, dups = ["%09d" % int(x) for x in rng.integers(0, 10 ** 9, 150000)], None
    vec = HipTfidfVectorizer(dtype=np.float32, ctx=ctx, **kw)
    pm = vec.prepare(master)
    sets = [pm] + ([vec.prepare(dups)] if dups is not None else [])
    vec.fit_prepared(sets)
    dA = vec.transform_prepared(pm)
    dB = dA if dups is None else vec.transform_prepared(sets[1])
    out = _multiply_both_ways(ctx, dA, dB, top_n, thr, monkeypatch)
    assert_csr_identical(out["1"][0], out["0"][0], family)
    t_pruned, t_exact = out["1"][1]["ms_spgemm_topn"], out["0"][1]["ms_spgemm_topn"]
    assert out["1"][1]["prune_rows"] > 0, family
    import os
    if not os.environ.get("SG_HIP_LIB"):        # (a diagnostic build of the library, e.g. with the loop watchdog, is not timed)
        assert t_pruned < 0.7 * t_exact, (family, t_pruned, t_exact)


def test_pruned_or_exact_is_decided_by_a_pilot_on_dense_vocabularies(ctx, monkeypatch):
    """2-grams: 875 terms, a row holds 2 % of the vocabulary -- the filter passes thousands of candidates per row and the
    exact kernel is the faster one; the library prices both from a pilot (three blocks of 512 rows).  Whatever it picks,
    the result is the oracle's."""
    from string_grouper_amd.vectorizer import HipTfidfVectorizer
    names = _names(40000, seed=31)
    (A_ref,), _, _ = O.tfidf_sklearn(names, [names], dtype=np.float32, ngram_size=2)
    C_ref = P.sp_matmul_topn_port(A_ref, A_ref.T, 10, 0.8, True, 8)
    vec = HipTfidfVectorizer(dtype=np.float32, ctx=ctx, ngram_size=2)
    p = vec.prepare(names)
    vec.fit_prepared([p])
    dA = vec.transform_prepared(p)
    post = ctx.postings_build(dA)
    picked = {}
    for label, env in (("pilot", {}), ("forced pruned", {"SG_PRUNE_PILOT": "0"}), ("forced self-join form", {"SG_PRUNE_PILOT": "0", "SG_SYM": "1"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        res = ctx.spgemm_topn(dA, post, 10, 0.8, True)
        picked[label] = ctx.stats()
        assert_csr_identical(res.to_scipy(), C_ref, label)
        res.free()
        for k in env:
            monkeypatch.delenv(k)
    assert picked["forced pruned"]["prune_rows"] > 0 and picked["forced self-join form"]["prune_symmetric"] == 1
    # the pilot's choice must be the cheaper of the two on this input
    t_pruned, t_pilot = picked["forced pruned"]["ms_spgemm_topn"], picked["pilot"]["ms_spgemm_topn"]
    assert t_pilot <= 1.25 * t_pruned + 0.5, (t_pilot, t_pruned)
    post.free()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("top_n,thr", [(10, 0.8), (64, 0.3), (5, 0.95), (20, 0.7), (3, 0.1), (1, 0.5), (64, 0.6)])
def test_pruned_multiply_equals_exact_and_oracle(ctx, mats, dtype, top_n, thr, monkeypatch):
    A = mats[dtype]
    dA = ctx.csr_from_scipy(A)
    out = _multiply_both_ways(ctx, dA, dA, top_n, thr, monkeypatch)
    st = out["1"][1]
    if thr >= 0.45:   # it did take the pruned kernel, and pruned
        assert st["prune_rows"] > 0 and st["prune_postings"] < st["macs"], st
    else:             # low thresholds pass too much through the filter: the library keeps the exact kernel
        assert st["prune_rows"] == 0, st
    assert out["0"][1]["prune_rows"] == 0
    what = f"{dtype.__name__} top_n={top_n} thr={thr}"
    assert_csr_identical(out["1"][0], out["0"][0], what + " pruned vs exact")
    assert_csr_identical(out["1"][0], P.sp_matmul_topn_port(A, A.T, top_n, thr, True, 8), what + " pruned vs oracle")


def test_low_thresholds_can_be_forced_through_the_pruned_kernel(ctx, mats, monkeypatch):
    A = mats[np.float32][:6000]
    dA = ctx.csr_from_scipy(A)
    for thr in (0.1, 0.3):
        out = _multiply_both_ways(ctx, dA, dA, 10, thr, monkeypatch, SG_PRUNE_MIN_THRESHOLD=0.05)
        assert out["1"][1]["prune_rows"] > 0
        assert_csr_identical(out["1"][0], out["0"][0], f"thr {thr}")
        assert_csr_identical(out["1"][0], P.sp_matmul_topn_port(A, A.T, 10, thr, True, 8), f"thr {thr}")


@pytest.mark.parametrize("env", [{"SG_PRUNE_DELTA": 0.02}, {"SG_PRUNE_DELTA": 0.35}, {"SG_PRUNE_FREQ": 0.0},
                                 {"SG_PRUNE_FREQ": 0.05}, {"SG_PRUNE_FREQ": 2.0}, {"SG_PRUNE_TILE": 11},
                                 {"SG_PRUNE_TILE": 13}])
def test_pruned_multiply_is_exact_for_every_tuning(ctx, mats, env, monkeypatch):
    """delta / the frequent-term share / the tile only move work between the filter and the exact scoring."""
    A = mats[np.float32][:8000]
    B = mats[np.float32][2000:20000]
    dA, dB = ctx.csr_from_scipy(A), ctx.csr_from_scipy(B)
    out = _multiply_both_ways(ctx, dA, dB, 10, 0.8, monkeypatch, **env)
    assert out["1"][1]["prune_rows"] > 0
    assert_csr_identical(out["1"][0], out["0"][0], str(env))
    assert_csr_identical(out["1"][0], P.sp_matmul_topn_port(A, B.T, 10, 0.8, True, 8), str(env))


def test_sorted_lists_are_as_fast_as_shuffled_ones(ctx, monkeypatch):
    """A sorted name list has its similar names side by side: a row's candidates pile up in a few column tiles and, in
    row order, the pruned multiply ran 2.6 x slower on 663 k sorted names than on the same names shuffled.  The index
    is therefore built over a fixed permutation of the right-hand rows (sg_postings.hip); nothing of it shows in the
    result -- rows, columns, the order and the cut of EQUAL scores (hubs of identical names, neighbours here) are the
    port's -- and the sorted list takes the time of the shuffled one."""
    import os
    names = _names(150000, 77)
    names_sorted = sorted(names)
    ms = {}
    for tag, lst in (("shuffled", names), ("sorted", names_sorted)):
        A = _tfidf(lst, np.float32)
        dA = ctx.csr_from_scipy(A)
        for permute in ("1", "0"):
            monkeypatch.setenv("SG_PERMUTE", permute)
            post = ctx.postings_build(dA)
            best = 1e9
            for _ in range(3):
                res = ctx.spgemm_topn(dA, post, 10, 0.8, True)
                st = ctx.stats()
                best = min(best, st["ms_spgemm_topn"])
                if _ == 0:
                    C = res.to_scipy()
                res.free()
            assert st["prune_symmetric"] == 1
            post.free()
            ms[tag, permute] = best
            if permute == "1" or tag == "sorted":
                assert_csr_identical(C, P.sp_matmul_topn_port(A, A.T, 10, 0.8, True, 16), f"{tag}, permutation {permute}")
        monkeypatch.delenv("SG_PERMUTE")
        dA.free()
    print("multiply ms:", ms)
    if not os.environ.get("SG_HIP_LIB"):
        assert ms["sorted", "1"] < 1.25 * ms["shuffled", "1"], ms      # (in row order: 2.6 x at 663 k, profiles/r02_sessionAD_*)


def test_pruned_multiply_rows_beyond_64_terms(ctx, monkeypatch):
    """Rows with 65 .. 128 distinct n-grams take the pruned kernel's second (wide) launch -- two staged terms per lane,
    the row's sorted terms searched instead of hashed -- in both forms; rows beyond that, or with more than 64 prefix
    terms, go to the exact kernel (and switch the self-join form off)."""
    rng = np.random.default_rng(5)
    letters = np.array(list("ABCDEFGHIJKLMNOPQRSTUVWXYZ "))
    base = _names(6000, 9)
    joined = [" ".join(base[4 * i:4 * i + 4]) for i in range(600)]               # ~100 characters: mostly 65 .. 128 n-grams
    medium = joined + [s[:-3] for s in joined[:200]] + [s + " LTD" for s in joined[200:300]]
    names = list(base[:3000]) + medium
    for dtype in (np.float32, np.float64):
        A = _tfidf(names, dtype)
        n_wide = int(((np.diff(A.indptr) > 64) & (np.diff(A.indptr) <= 128)).sum())
        assert n_wide >= 500 and (np.diff(A.indptr) > 128).sum() == 0
        dA = ctx.csr_from_scipy(A)
        out = _multiply_both_ways(ctx, dA, dA, 5, 0.6, monkeypatch)
        st = out["1"][1]
        assert st["exact_rows"] < n_wide // 10 and st["prune_rows"] >= len(names) - 50, st      # the wide launch took them
        C_ref = P.sp_matmul_topn_port(A, A.T, 5, 0.6, True, 8)
        assert_csr_identical(out["1"][0], C_ref, "one-sided with wide rows")
        assert_csr_identical(out["sym"][0], C_ref, "self-join form with wide rows")
        assert out["sym"][1]["prune_symmetric"] == 1
        # without the wide launch the same rows go to the exact kernel: same bits
        monkeypatch.setenv("SG_PRUNE_WIDE", "0")
        out0 = _multiply_both_ways(ctx, dA, dA, 5, 0.6, monkeypatch)
        monkeypatch.delenv("SG_PRUNE_WIDE")
        assert out0["1"][1]["exact_rows"] >= n_wide
        assert_csr_identical(out0["1"][0], C_ref, "wide launch off")
    # rows beyond 128 terms (random letters: every n-gram distinct) are the exact kernel's -- in the self-join form
    # through its self-join launch INSIDE the pass (pairs j <= i, mirrored ones into the pair list): such rows match
    # rows of every kind here, before and behind them (shorter cuts of themselves that the wide launch takes, other
    # long rows, a hub of copies whose top-n is cut in the merge)
    long_names = ["".join(rng.choice(letters, 150)) for _ in range(40)]
    names = (list(base[:1500]) + [s[:125] for s in long_names[:20]] + long_names + [s[:140] + "X" for s in long_names]
             + [s[:126] for s in long_names[10:30]] + [long_names[5]] * 9 + list(base[1500:3000]) + medium[:100]
             + [long_names[7][:148]] * 3)
    for dtype in (np.float32, np.float64):
        A = _tfidf(names, dtype)
        n_long = int((np.diff(A.indptr) > 128).sum())
        assert n_long >= 90
        dA = ctx.csr_from_scipy(A)
        for top_n, thr in ((5, 0.6), (10, 0.8)):
            out = _multiply_both_ways(ctx, dA, dA, top_n, thr, monkeypatch)
            st = out["1"][1]
            assert st["exact_rows"] >= n_long and st["prune_rows"] > 0, st
            sym = out["sym"][1]
            assert sym["prune_symmetric"] == 1 and sym["exact_rows"] >= n_long, sym   # the form no longer stands down
            C_ref = P.sp_matmul_topn_port(A, A.T, top_n, thr, True, 8)
            assert (np.diff(C_ref.indptr)[np.diff(A.indptr) > 128] > 1).sum() >= 80   # the long rows do have matches
            assert_csr_identical(out["1"][0], C_ref, "one-sided with long rows")
            assert_csr_identical(out["sym"][0], C_ref, "self-join form with long rows")


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_selfjoin_form_over_row_ranges_equals_the_whole(ctx, dtype):
    """The multi-GPU form of the self-join (sg_selfjoin_range / sg_selfjoin_merge) with the ranks played one after the
    other on this GPU: every range scores its pairs (i, j <= i), the pair lists are concatenated, every range merges
    the pairs that point into it -- the rows put together must equal the one-GPU result and the port, bit for bit, for
    one, two, three and five ranges (hubs of duplicates: rows whose top-n is cut inside the merge; rows of 65 .. 128
    terms: the wide launch inside a range)."""
    import torch
    from string_grouper_amd import distributed as D
    from string_grouper_amd.vectorizer import HipTfidfVectorizer
    base = list(_names(9000, 21))
    rng = np.random.default_rng(4)
    wide = [" ".join(rng.choice(base, 5)) for _ in range(60)]
    names = base + [base[3]] * 150 + wide + [base[11] + " CO"] * 90 + [w + " X" for w in wide[:20]]
    A = _tfidf(names, dtype)
    assert int((np.diff(A.indptr) > 64).sum()) > 20
    dA = ctx.csr_from_scipy(A)
    want = P.sp_matmul_topn_port(A, A.T, 10, 0.75, True, 8)
    ops = D.HipOps(ctx, lambda: HipTfidfVectorizer(dtype=dtype, ctx=ctx))
    n = len(names)

    def stacked(rows, blk_orig_of):
        """the ranks' blocks put together; with the index over the library's row permutation the ranges are ranges of
        POSITIONS and block row p is row orig_of[p]"""
        C = sp.vstack(rows).tocsr()
        if blk_orig_of is None:
            return C
        inverse = np.argsort(blk_orig_of.cpu().numpy(), kind="stable")
        return C[inverse]

    ctx.set_option("SG_COLLAPSE", "0")            # (an index over every row first; over groups of identical rows below)
    for permute in (False, True):
        post = ctx.postings_build(dA, permute=permute)
        assert ctx.postings_rows(post) == (n, n, 0)
        for world in (1, 2, 3, 5):
            bounds = D.selfjoin_row_ranges(n, world)
            assert bounds[0] == 0 and bounds[-1] == n
            parts = [ops.selfjoin_range(dA, post, 10, 0.75, int(bounds[r]), int(bounds[r + 1])) for r in range(world)]
            assert all(p is not None for p in parts)
            pairs_all = torch.cat([ops.selfjoin_pairs(p).clone() for p in parts])
            assert pairs_all.numel() % parts[0]["words"] == 0 and pairs_all.numel() > 0
            rows, orig_of = [], None
            for r in range(world):
                blk = ops.selfjoin_merge(parts[r], pairs_all, int(bounds[r]), int(bounds[r + 1]))
                assert (blk.orig_of is not None) == permute
                orig_of = blk.orig_of
                rows.append(blk.to_scipy())
                blk.free()
            assert_csr_identical(stacked(rows, orig_of), want, f"{world} ranges, permutation {permute}")
        post.free()
    # ---- the index over one representative per group of identical rows (the default when enough rows repeat): the
    #      ranges are ranges of the groups' positions, a rank's block holds groups, the gathered blocks are expanded
    ctx.set_option("SG_COLLAPSE", "1")
    for permute in (False, True):
        post = ctx.postings_build(dA, permute=permute)
        n_groups, n_rows, p_gid = ctx.postings_rows(post)
        assert p_gid and n_rows == n and n_groups == len({(A.indices[a:b].tobytes(), A.data[a:b].tobytes())
                                                for a, b in zip(A.indptr[:-1], A.indptr[1:])}) < n - 230
        for world in (1, 2, 5):
            bounds = D.selfjoin_row_ranges(n_groups, world)
            parts = [ops.selfjoin_range(dA, post, 10, 0.75, int(bounds[r]), int(bounds[r + 1])) for r in range(world)]
            assert all(p is not None for p in parts)
            pairs_all = torch.cat([ops.selfjoin_pairs(p).clone() for p in parts])
            blocks = [ops.selfjoin_merge(parts[r], pairs_all, int(bounds[r]), int(bounds[r + 1])) for r in range(world)]
            # a rank's block: the rows that are members of its groups, expanded on the rank (sg_topn_expand_groups)
            assert all(b.row_ids is not None and b.orig_of is None and b.dims()[0] == b.row_ids.numel() for b in blocks)
            ids = torch.cat([b.row_ids for b in blocks]).to(torch.int64)
            assert ids.numel() == n and torch.equal(torch.sort(ids).values.cpu(), torch.arange(n))
            # what gather_topn does with the ranks' blocks: concatenate, put every row where its number says
            cols, vals, counts = (torch.cat(t) for t in zip(*[ops.topn_tensors(b) for b in blocks]))
            cols, vals, counts = (torch.empty_like(t).index_copy_(0, ids, t).cpu().numpy() for t in (cols, vals, counts))
            mask = np.arange(cols.shape[1])[None, :] < counts[:, None]
            got = sp.csr_matrix((vals[mask], cols[mask], np.concatenate([[0], np.cumsum(counts)])), shape=(n, n))
            assert_csr_identical(got, want, f"{world} ranges of groups, permutation {permute}")
            for b in blocks:
                b.free()
        post.free()
    # ---- INTERLEAVED shares (the driver's default: rank r scores every world-th position from the top, sg_selfjoin_range's
    #      row_step): blocks list their rows; plain index and index over groups, with and without the row permutation
    for collapse in ("0", "1"):
        ctx.set_option("SG_COLLAPSE", collapse)
        for permute in (False, True):
            post = ctx.postings_build(dA, permute=permute)
            n_index = ctx.postings_rows(post)[0]
            assert (n_index < n) == (collapse == "1")
            for world in (2, 3, 7):
                shares = [D.selfjoin_share(n_index, r, world) for r in range(world)]
                assert all(sh[2] == world for sh in shares)
                parts = [ops.selfjoin_range(dA, post, 10, 0.75, *shares[r]) for r in range(world)]
                assert all(p is not None for p in parts)
                pairs_all = torch.cat([ops.selfjoin_pairs(p).clone() for p in parts])
                blocks = [ops.selfjoin_merge(parts[r], pairs_all, *shares[r]) for r in range(world)]
                assert all(b.row_ids is not None and b.dims()[0] == b.row_ids.numel() for b in blocks)
                ids = torch.cat([b.row_ids for b in blocks]).to(torch.int64)
                assert ids.numel() == n and torch.equal(torch.sort(ids).values.cpu(), torch.arange(n))
                cols, vals, counts = (torch.cat(t) for t in zip(*[ops.topn_tensors(b) for b in blocks]))
                cols, vals, counts = (torch.empty_like(t).index_copy_(0, ids, t).cpu().numpy() for t in (cols, vals, counts))
                mask = np.arange(cols.shape[1])[None, :] < counts[:, None]
                got = sp.csr_matrix((vals[mask], cols[mask], np.concatenate([[0], np.cumsum(counts)])), shape=(n, n))
                assert_csr_identical(got, want, f"{world} interleaved shares, groups {collapse}, permutation {permute}")
                for b in blocks:
                    b.free()
            post.free()
    ctx.set_option("SG_COLLAPSE", "0")
    # rows for the exact kernel (more than 128 terms) are scored inside the range's pass by that kernel's self-join launch
    extra = ["".join(rng.choice(list("ABCDEFGHIJKLMNOPQRSTUVWXYZ"), 180)) for _ in range(3)]
    long_names = names[:4000] + extra + names[4000:] + [extra[1][:170], extra[2]]
    AL = _tfidf(long_names, dtype)
    dL = ctx.csr_from_scipy(AL)
    postL = ctx.postings_build(dL)                     # (over the row permutation: ranges of positions)
    wantL = P.sp_matmul_topn_port(AL, AL.T, 10, 0.75, True, 8)
    nL = len(long_names)
    for world in (1, 3):
        bounds = D.selfjoin_row_ranges(nL, world)
        parts = [ops.selfjoin_range(dL, postL, 10, 0.75, int(bounds[r]), int(bounds[r + 1])) for r in range(world)]
        assert all(p is not None for p in parts)
        pairs_all = torch.cat([ops.selfjoin_pairs(p).clone() for p in parts])
        rows, orig_of = [], None
        for r in range(world):
            blk = ops.selfjoin_merge(parts[r], pairs_all, int(bounds[r]), int(bounds[r + 1]))
            orig_of = blk.orig_of
            rows.append(blk.to_scipy())
            blk.free()
        assert orig_of is not None
        assert_csr_identical(stacked(rows, orig_of), wantL, f"{world} ranges with rows for the exact kernel")
    postL.free(); dL.free(); dA.free()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_rows_worked_off_in_parts_give_the_same_rows(ctx, dtype):
    """Stream + self-join form: rows that need many rounds are set aside and worked off in sixteen parts each by a second
    launch (a part hands its matches to the pair list, addressed to its own row).  With the bar at 2 rounds nearly every
    row of more than one super-tile goes that way: the result must not change by a bit -- whole matrix and ranges --
    and equals the port's."""
    import torch
    from string_grouper_amd import distributed as D
    from string_grouper_amd.vectorizer import HipTfidfVectorizer
    base = list(_names(90000, 33))
    names = base + [base[7]] * 120 + [base[70001] + " CO"] * 80 + ["A INC", "B INC", "AB INC", "INC"]
    A = _tfidf(names, dtype)
    dA = ctx.csr_from_scipy(A)
    n = len(names)
    want = P.sp_matmul_topn_port(A, A.T, 10, 0.8, True, 16)
    ctx.set_option("SG_COLLAPSE", "0")
    ctx.set_option("SG_SYM", "1")
    ops = D.HipOps(ctx, lambda: HipTfidfVectorizer(dtype=dtype, ctx=ctx))
    for bar in ("0", "2", "40"):
        ctx.set_option("SG_HEAVY_ROUNDS", bar)
        post = ctx.postings_build(dA)
        res = ctx.spgemm_topn(dA, post, 10, 0.8, True)
        assert ctx.stats()["prune_symmetric"] == 1
        assert_csr_identical(res.to_scipy(), want, f"whole matrix, rows of >= {bar} rounds in parts")
        res.free()
        bounds = D.selfjoin_row_ranges(n, 3)
        parts = [ops.selfjoin_range(dA, post, 10, 0.8, int(bounds[r]), int(bounds[r + 1])) for r in range(3)]
        assert all(p is not None for p in parts)
        pairs_all = torch.cat([ops.selfjoin_pairs(p).clone() for p in parts])
        rows, orig_of = [], None
        for r in range(3):
            blk = ops.selfjoin_merge(parts[r], pairs_all, int(bounds[r]), int(bounds[r + 1]))
            orig_of = blk.orig_of
            rows.append(blk.to_scipy())
            blk.free()
        C = sp.vstack(rows).tocsr()[np.argsort(orig_of.cpu().numpy(), kind="stable")]
        assert_csr_identical(C, want, f"three ranges, rows of >= {bar} rounds in parts")
        post.free()
    dA.free()


def test_selfjoin_form_with_a_pair_list_that_is_too_small_falls_back(ctx, monkeypatch):
    """The self-join form collects the mirrored pairs in a list of bounded size (chunks handed to the waves); hubs of
    duplicates can outgrow it.  Then nothing of that pass may survive: the one-sided form runs and the result is the
    same, bit for bit."""
    base = list(_names(6000, 5))
    names = base + [base[3]] * 400 + [base[9] + " LLC"] * 300
    A = _tfidf(names, np.float32)
    dA = ctx.csr_from_scipy(A)
    ref = _multiply_both_ways(ctx, dA, dA, 10, 0.8, monkeypatch)
    assert ref["sym"][1]["prune_symmetric"] == 1
    small = _multiply_both_ways(ctx, dA, dA, 10, 0.8, monkeypatch, expect_sym=False, SG_SYM_PAIR_CAP=512)     # two chunks
    monkeypatch.delenv("SG_SYM_PAIR_CAP")
    assert small["sym"][1]["prune_symmetric"] == 0 and small["sym"][1]["prune_rows"] > 0
    assert_csr_identical(small["sym"][0], ref["0"][0], "fallback after a full pair list")
    assert_csr_identical(ref["sym"][0], P.sp_matmul_topn_port(A, A.T, 10, 0.8, True, 8))


def test_pruned_multiply_with_hubs_of_duplicates(ctx, monkeypatch):
    """Hundreds of identical / near-identical names: the survivor buffer drains many times per row and
    the top-n cut falls inside a block of equal scores."""
    base = list(_names(2000, 3))
    names = base + [base[7]] * 300 + [base[11] + " INC"] * 200 + [base[11]] * 150
    for dtype in (np.float32, np.float64):
        A = _tfidf(names, dtype)
        dA = ctx.csr_from_scipy(A)
        for top_n, thr in ((64, 0.5), (10, 0.9)):
            out = _multiply_both_ways(ctx, dA, dA, top_n, thr, monkeypatch)
            assert out["1"][1]["prune_rows"] > 0
            assert_csr_identical(out["1"][0], out["0"][0], f"{dtype.__name__} {top_n} {thr}")
            assert_csr_identical(out["1"][0], P.sp_matmul_topn_port(A, A.T, top_n, thr, True, 8))


def test_matrices_that_are_not_cosine_like_take_the_exact_kernel(ctx, mats, monkeypatch):
    """Row norms above 1, negative values or unsorted rows: no pruning (its bounds would not hold)."""
    monkeypatch.delenv("SG_PRUNE", raising=False)
    A = mats[np.float32][:4000].copy()
    scaled = A * np.float32(1.5)
    negative = A.copy()
    negative.data[::7] *= np.float32(-1.0)
    for M, thr in ((scaled, 0.8), (negative, 0.5)):
        M = sp.csr_matrix(M)
        d = ctx.csr_from_scipy(M)
        post = ctx.postings_build(d)
        res = ctx.spgemm_topn(d, post, 10, thr, True)
        st = ctx.stats()
        assert st["prune_rows"] == 0, st
        assert_csr_identical(res.to_scipy(), P.sp_matmul_topn_port(M, M.T, 10, thr, True, 8))
        res.free()
        post.free()
    # a cosine-like right-hand side with a left-hand side that is not: exact kernel as well
    dB, dA = ctx.csr_from_scipy(A), ctx.csr_from_scipy(sp.csr_matrix(scaled))
    post = ctx.postings_build(dB)
    res = ctx.spgemm_topn(dA, post, 10, 0.8, True)
    assert ctx.stats()["prune_rows"] == 0
    assert_csr_identical(res.to_scipy(), P.sp_matmul_topn_port(sp.csr_matrix(scaled), A.T, 10, 0.8, True, 8))
    res.free()
    post.free()


# ------------------------------------------------------------------------------------------------
# K7 / K8: the reductions over the match list on the device (best master per duplicate, group
# representatives) against the host formulation of the same reference code on the same match list.
def _groups_both_ways(sg):
    assert "_device_matches" in sg.__dict__, "fit() did not keep the match list on the device"
    on_device = sg.get_groups()
    sg._drop_device_matches()
    on_host = sg.get_groups()
    return on_device, on_host


def _assert_same_frames(a, b):
    import pandas as pd
    assert type(a) is type(b)
    if isinstance(a, pd.Series):
        pd.testing.assert_series_equal(a, b)
    else:
        pd.testing.assert_frame_equal(a, b)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("group_rep", ["centroid", "first"])
def test_group_representatives_on_device_equal_host(ctx, dtype, group_rep):
    import pandas as pd
    import string_grouper_amd as sga
    import string_grouper_amd.engine as E
    E.set_engine(E.HipEngine(ctx))
    base = list(_names(6000, 21))
    # a long chain (every name one edit away from the next: a component with a large diameter), hubs of
    # duplicates (equal row sums: the tie goes to the lowest index) and ordinary names
    chain = ["ALPHABETAGAMMADELTAEPSILONZETA HOLDINGS INTERNATIONAL CORPORATION"]
    for i in range(120):
        s = chain[-1]
        chain.append(s[:5 + (i % 40)] + "X" + s[6 + (i % 40):])
    hub = [base[9] + " " + str(i % 7) for i in range(300)]        # rows with > 128 entries of unequal similarity
    names = base + chain + [base[3]] * 40 + [base[5] + " INC"] * 25 + hub
    s = pd.Series(names)
    sg = sga.StringGrouper(s, min_similarity=0.8, tfidf_matrix_dtype=dtype, group_rep=group_rep,
                           max_n_matches=400).fit()
    dev, host = _groups_both_ways(sg)
    _assert_same_frames(dev, host)
    reps = dev if isinstance(dev, pd.Series) else dev.iloc[:, -1]
    assert reps.nunique() < len(names)         # it did group something
    # with ids
    ids = pd.Series(np.arange(len(names)) * 7)
    sg = sga.StringGrouper(s, master_id=ids, min_similarity=0.7, tfidf_matrix_dtype=dtype, group_rep=group_rep).fit()
    dev, host = _groups_both_ways(sg)
    _assert_same_frames(dev, host)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_best_master_per_duplicate_on_device_equals_host(ctx, dtype):
    import pandas as pd
    import string_grouper_amd as sga
    import string_grouper_amd.engine as E
    from string_grouper_amd.synth import synth_names
    E.set_engine(E.HipEngine(ctx))
    master = list(_names(8000, 31))
    master = master + master[:500]                      # exact duplicates inside the master: ties at the maximum
    dupes = list(synth_names(5000, seed=32, perturb_of=np.asarray(master, dtype=object), perturb_frac=0.6))
    m, d = pd.Series(master), pd.Series(dupes)
    for kw in (dict(min_similarity=0.8), dict(min_similarity=0.6, max_n_matches=5),
               dict(min_similarity=0.7, ignore_index=True)):
        sg = sga.StringGrouper(m, d, tfidf_matrix_dtype=dtype, **kw).fit()
        dev, host = _groups_both_ways(sg)
        _assert_same_frames(dev, host)
    # and through the public function, with ids
    mid, did = pd.Series(np.arange(len(master))), pd.Series(np.arange(len(dupes)) + 10 ** 6)
    out = sga.match_most_similar(m, d, master_id=mid, duplicates_id=did, min_similarity=0.8, tfidf_matrix_dtype=dtype)
    sg = sga.StringGrouper(m, d, mid, did, min_similarity=0.8, tfidf_matrix_dtype=dtype, max_n_matches=1).fit()
    sg._drop_device_matches()          # (match_most_similar asks for one match per master row, string_grouper.py:120)
    _assert_same_frames(out, sg.get_groups())


def test_editing_the_match_list_drops_the_device_copy(ctx):
    import pandas as pd
    import string_grouper_amd as sga
    import string_grouper_amd.engine as E
    E.set_engine(E.HipEngine(ctx))
    s = pd.Series(list(_names(500, 41)))
    sg = sga.StringGrouper(s, min_similarity=0.8).fit()
    assert "_device_matches" in sg.__dict__
    sg = sg.add_match(s.iloc[1], s.iloc[2])
    assert "_device_matches" not in sg.__dict__
    groups = sg.get_groups()
    reps = groups if isinstance(groups, pd.Series) else groups.iloc[:, -1]
    assert reps.iloc[1] == reps.iloc[2]


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_rowwise_dot_on_device_equals_scipy(ctx, dtype):
    """K9 (StringGrouper.dot / compute_pairwise_similarities, string_grouper.py:433-440) against scipy's
    multiply(...).sum(axis=1) on the same matrices: bit-identical, incl. long rows (pairwise summation)."""
    import pandas as pd
    import string_grouper_amd as sga
    import string_grouper_amd.engine as E
    from string_grouper_amd.synth import synth_names
    E.set_engine(E.HipEngine(ctx))
    rng = np.random.default_rng(3)
    letters = np.array(list("ABCDEFGHIJ "))
    left = list(_names(4000, 51)) + ["".join(rng.choice(letters, 400)) for _ in range(50)] + ["", "AB", "XYZ"]
    other = list(synth_names(4000, seed=52))
    right = [(left[i] + " INC" if i % 2 else left[i][:-1]) if i % 3 else other[i] for i in range(4000)]   # row-aligned edits
    right += [s[:350] + "Q" + s[351:] for s in left[4000:4050]] + ["", "AB", "XYW"]
    s1, s2 = pd.Series(left), pd.Series(right)
    got = sga.compute_pairwise_similarities(s1, s2, tfidf_matrix_dtype=dtype)
    A = _tfidf(left + right, dtype)          # same vocabulary / idf as fit on master + duplicates
    a, b = A[:len(left)], A[len(left):]
    want = np.asarray(a.multiply(b).sum(axis=1)).squeeze(axis=1)
    assert got.dtype == want.dtype
    np.testing.assert_array_equal(got.to_numpy(), want)
    assert (got.to_numpy() > 0.5).sum() > 1000


def test_pruned_multiply_with_unsorted_output_and_row_blocks(ctx, mats, monkeypatch):
    """sort=False (rows re-ordered by column afterwards) and a left matrix that is a row-block view
    (how the multi-GPU path hands out work) through the pruned kernel."""
    monkeypatch.delenv("SG_PRUNE", raising=False)
    A = mats[np.float32]
    dA = ctx.csr_from_scipy(A)
    post = ctx.postings_build(dA)
    for lo, hi in ((0, 20000), (5000, 12345), (19990, 20000)):
        blk = dA.row_block(lo, hi)
        for sort in (True, False):
            res = ctx.spgemm_topn(blk, post, 10, 0.7, sort)
            st = ctx.stats()
            assert st["prune_rows"] > 0
            assert_csr_identical(res.to_scipy(), P.sp_matmul_topn_port(A[lo:hi], A.T, 10, 0.7, sort, 8), f"{lo}:{hi} sort={sort}")
            res.free()
        blk.free()
    post.free()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_postings_built_with_lds_and_with_global_counters_give_the_same_multiply(ctx, mats, dtype, monkeypatch):
    """K3 keeps a tile's counters in LDS when the vocabulary fits and in global memory otherwise; the
    order of the entries inside a posting segment differs, the multiply's result must not."""
    A = mats[dtype][:9000]
    B = mats[dtype][4000:20000]
    dA, dB = ctx.csr_from_scipy(A), ctx.csr_from_scipy(B)
    want = P.sp_matmul_topn_port(A, B.T, 10, 0.6, True, 8)
    for lds in ("1", "0"):
        monkeypatch.setenv("SG_POSTINGS_LDS", lds)
        for prune in ("1", "0"):
            monkeypatch.setenv("SG_PRUNE", prune)
            post = ctx.postings_build(dB)
            res = ctx.spgemm_topn(dA, post, 10, 0.6, True)
            assert_csr_identical(res.to_scipy(), want, f"{dtype.__name__} lds={lds} prune={prune}")
            res.free()
            post.free()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("top_n,thr", [(10, 0.8), (5, 0.6)])
def test_hip_equals_the_canonical_tie_rule_exactly_and_the_arrival_rule_tie_aware(ctx, dtype, top_n, thr):
    """Hubs of identical names larger than top_n: the HIP path equals the canonical port bit for bit, and differs from
    the arrival-order variant (tests/test_tie_rules.py) only in which entries AT a row's cut score it keeps -- what a
    user coming from the real wheel can see change (README, "ties")."""
    from string_grouper_amd.sparse_dot_topn import sp_matmul_topn
    from tests.test_tie_rules import hub_names
    names = hub_names(30000, 11)
    A = _tfidf(names, dtype)
    C_dev = sp_matmul_topn(A, A.T, top_n, thr, sort=True, ctx=ctx)
    assert_csr_identical(C_dev, P.sp_matmul_topn_port(A, A.T, top_n, thr, True, 8), "canonical rule")
    arrival = P.sp_matmul_topn_port(A, A.T, top_n, thr, True, 8, tie_rule=1)
    assert (C_dev != arrival).nnz > 0
    assert O.compare_tie_aware(C_dev, arrival, top_n) == []


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("order", ["shuffled", "sorted"])
def test_identical_rows_are_collapsed_and_the_result_is_the_ports(ctx, dtype, order, monkeypatch):
    """Hubs of identical names (3 000, 700, 40 and pairs) with near-duplicates around them: the index is built over one
    representative per group (sg_collapse.hip), the multiply runs on groups and is expanded -- self-join and one-sided,
    top_n below and above a hub's size, top_n above the register list (the plain index on demand), collapse forced off --
    always the port's result, bit for bit, ties at the cut included (lowest columns win)."""
    from string_grouper_amd.sparse_dot_topn import sp_matmul_topn
    rng = np.random.default_rng(21)
    names = _names(20000, seed=77)
    for hub, size in (("ACME HOLDINGS INC", 3000), ("ZENITH CAPITAL PARTNERS LP", 700), ("OMEGA TRUST", 40)):
        for at in rng.choice(len(names), size, replace=False):
            names[at] = hub
    for at in rng.choice(len(names), 300, replace=False):
        names[at] = "ACME HOLDINGS INC" + rng.choice([".", " 2", "ORPORATED", " LLC"])
    for at in rng.choice(len(names) - 1, 500, replace=False):
        names[at + 1] = names[at]                              # pairs
    # rows the pruned kernel hands on (more than 128 terms: the exact kernel's rows), some of them identical too
    longs = ["".join(rng.choice(list("ABCDEFGHIJKLMNOPQRSTUVWXYZ "), 170)) for _ in range(4)]
    for k, at in enumerate(rng.choice(len(names), 12, replace=False)):
        names[at] = longs[k % 4] if k < 10 else longs[0][:160]
    if order == "sorted":
        names = sorted(names)
    A = _tfidf(names, dtype)
    assert int((np.diff(A.indptr) > 128).sum()) >= 10
    for top_n, thr in ((10, 0.8), (5, 0.6), (64, 0.8), (100, 0.8)):
        want = P.sp_matmul_topn_port(A, A.T, top_n, thr, True, 8)
        for collapse in ("1", "0"):
            monkeypatch.setenv("SG_COLLAPSE", collapse)
            got = sp_matmul_topn(A, A.T, top_n, thr, sort=True, ctx=ctx)
            assert_csr_identical(got, want, f"self-join top_n={top_n} thr={thr} collapse={collapse} {order}")
    # ... and in the self-join form (forced at this size): groups + rows for the exact kernel's self-join launch + the
    # postings proper written on demand
    monkeypatch.setenv("SG_COLLAPSE", "1")
    monkeypatch.setenv("SG_SYM", "1")
    dA = ctx.csr_from_scipy(A)               # (the form needs the left matrix to BE the matrix of the index)
    post = ctx.postings_build(dA)
    res = ctx.spgemm_topn(dA, post, 10, 0.8, True)
    st = ctx.stats()
    assert st["prune_symmetric"] == 1 and st["exact_rows"] >= 3 and st["prune_rows"] < A.shape[0] - 4000, st
    assert_csr_identical(res.to_scipy(), P.sp_matmul_topn_port(A, A.T, 10, 0.8, True, 8), f"self-join form on groups, {order}")
    for h in (res, post, dA):
        h.free()
    monkeypatch.delenv("SG_SYM")
    monkeypatch.setenv("SG_COLLAPSE", "1")
    left = A[1000:7000]
    assert_csr_identical(sp_matmul_topn(left, A.T, 10, 0.8, sort=True, ctx=ctx), P.sp_matmul_topn_port(left, A.T, 10, 0.8, True, 8),
                         "one-sided, columns expanded")
    assert_csr_identical(sp_matmul_topn(left, A.T, 7, 0.5, sort=False, ctx=ctx), P.sp_matmul_topn_port(left, A.T, 7, 0.5, False, 8),
                         "one-sided, sorted by column")
    monkeypatch.delenv("SG_COLLAPSE")
    # the public API on a list with hubs: groups and frames as on the uncollapsed path (the fuzz and fixture tests cover
    # the rest; here the hubs are larger than max_n_matches)
    import pandas as pd
    import string_grouper_amd as sga
    import string_grouper_amd.engine as E
    old = E._engine
    try:
        E.set_engine(E.HipEngine(ctx))
        s = pd.Series(names)
        monkeypatch.setenv("SG_COLLAPSE", "0")
        want_g = sga.group_similar_strings(s, min_similarity=0.8, tfidf_matrix_dtype=dtype)
        want_m = sga.match_strings(s, min_similarity=0.8, max_n_matches=8, tfidf_matrix_dtype=dtype)
        monkeypatch.setenv("SG_COLLAPSE", "1")
        pd.testing.assert_frame_equal(pd.DataFrame(sga.group_similar_strings(s, min_similarity=0.8, tfidf_matrix_dtype=dtype)),
                                      pd.DataFrame(want_g))
        pd.testing.assert_frame_equal(sga.match_strings(s, min_similarity=0.8, max_n_matches=8, tfidf_matrix_dtype=dtype), want_m)
    finally:
        E.set_engine(old)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_identical_left_rows_of_a_one_sided_product_are_multiplied_once(ctx, dtype, monkeypatch):
    """master x duplicates with a master list that repeats itself (BASELINE.json configs[4]: a fifth of 10 M names): the rows
    of the LEFT matrix are grouped (round 4), the representatives multiplied, every row gets its group's result row.  Above
    the automatic bar (65 536 rows) and forced below it; against two right-hand sides in turn (the groups are kept with
    the matrix); sorted by score and by column; with the right-hand side's own grouping on and off; top_n above the
    register list -- always the port's rows, bit for bit."""
    rng = np.random.default_rng(5)
    base = _names(60000, seed=91)
    master = base + [base[i] for i in rng.integers(0, len(base), 14000)] + ["NORTHWIND TRADERS LLC"] * 900
    master = [master[i] for i in rng.permutation(len(master))]
    from string_grouper_amd.synth import synth_names
    d1 = synth_names(9000, 92, perturb_of=base, perturb_frac=0.5) + ["NORTHWIND TRADERS LLC", "NORTHWIND TRADERS"]
    d2 = synth_names(12000, 93, perturb_of=base, perturb_frac=0.6) + [d1[7]] * 500          # (a right-hand side that repeats, too)
    (M, D1, D2), _, _ = O.tfidf_sklearn(master + d1 + d2, [master, d1, d2], dtype=dtype)
    for variant in ("default", "row order", "left off"):
        if variant == "row order":
            monkeypatch.setenv("SG_PERMUTE", "0")
            monkeypatch.setenv("SG_COLLAPSE_LEFT", "1")
        elif variant == "left off":
            monkeypatch.delenv("SG_PERMUTE")
            monkeypatch.setenv("SG_COLLAPSE_LEFT", "0")
        dM = ctx.csr_from_scipy(M)                # (the decision is kept with the matrix: a fresh one per variant)
        for D, top_n, thr, sort in ((D1, 20, 0.7, True), (D2, 20, 0.7, True), (D2, 5, 0.8, False), (D1, 100, 0.75, True)):
            dD = ctx.csr_from_scipy(D)
            post = ctx.postings_build(dD)
            res = ctx.spgemm_topn(dM, post, top_n, thr, sort)
            st = ctx.stats()
            assert_csr_identical(res.to_scipy(), P.sp_matmul_topn_port(M, D.T, top_n, thr, sort, 8),
                                 f"top_n={top_n} thr={thr} sort={sort} {variant}")
            if top_n <= 64:
                if variant == "left off":
                    assert st["prune_rows"] > 70000, st
                else:
                    assert 0 < st["prune_rows"] <= 60001, st      # the representatives, not the 74 900 rows
            assert st["out_nnz"] == res.to_scipy().nnz, st          # (entries kept are counted on ALL rows)
            for h in (res, post, dD):
                h.free()
        dM.free()
    monkeypatch.delenv("SG_COLLAPSE_LEFT")
    dM = ctx.csr_from_scipy(M)
    # a small left matrix: grouped only when forced; a row block (a view) groups on its own
    small = M[:9000]
    dS = ctx.csr_from_scipy(small)
    dD = ctx.csr_from_scipy(D1)
    post = ctx.postings_build(dD)
    for left in ("0", "1"):
        monkeypatch.setenv("SG_COLLAPSE_LEFT", left)
        for mat, ref in ((dS, small), (dM.row_block(30000, 50000), M[30000:50000])):
            res = ctx.spgemm_topn(mat, post, 20, 0.7, True)
            assert_csr_identical(res.to_scipy(), P.sp_matmul_topn_port(ref, D1.T, 20, 0.7, True, 8), f"small / view, left grouping {left}")
            res.free()
        dS.free()
        dS = ctx.csr_from_scipy(small)            # (a fresh object: the decision is kept with the matrix)
    for h in (post, dD, dS, dM):
        h.free()


def test_a_row_block_of_a_matrix_that_was_multiplied_groups_its_own_rows(ctx, monkeypatch):
    """a view took a COPY of the parent's groups of identical left rows -- indexed by the parent's
    rows, so rows r0.. of a view received the results of parent rows 0.. -- and freed them under the parent.  The parent is
    multiplied first (its groups are made), then row blocks with r0 > 0 are multiplied, freed, and the parent is
    multiplied and freed again: always the port's rows."""
    rng = np.random.default_rng(11)
    base = _names(50000, seed=77)
    master = base + [base[i] for i in rng.integers(0, len(base), 20000)]          # 70 000 rows, 2 in 7 repeat
    master = [master[i] for i in rng.permutation(len(master))]
    from string_grouper_amd.synth import synth_names
    dup = synth_names(8000, 78, perturb_of=base, perturb_frac=0.5)
    (M, D), _, _ = O.tfidf_sklearn(master + dup, [master, dup], dtype=np.float32)
    monkeypatch.setenv("SG_COLLAPSE_LEFT", "1")
    dM, dD = ctx.csr_from_scipy(M), ctx.csr_from_scipy(D)
    post = ctx.postings_build(dD)
    want = P.sp_matmul_topn_port(M, D.T, 10, 0.7, True, 8)
    res = ctx.spgemm_topn(dM, post, 10, 0.7, True)                  # the parent's groups exist from here on
    assert_csr_identical(res.to_scipy(), want, "parent, first multiply")
    res.free()
    for r0, r1 in ((20000, 45000), (1, 70000), (69000, 70000)):
        view = dM.row_block(r0, r1)
        res = ctx.spgemm_topn(view, post, 10, 0.7, True)
        assert_csr_identical(res.to_scipy(), want[r0:r1], f"view [{r0}, {r1}) of a parent with groups")
        res.free()
        view.free()                                                  # must not take the parent's groups with it
    res = ctx.spgemm_topn(dM, post, 10, 0.7, True)
    assert_csr_identical(res.to_scipy(), want, "parent, after its views were freed")
    for h in (res, post, dD, dM):
        h.free()


@pytest.mark.timeout(900)
def test_headline_663k_at_the_reference_defaults_fp64_top20_every_row_equals_sklearn_and_the_port(ctx):
    """The reference's DEFAULTS (string_grouper.py:18,20: tfidf_matrix_dtype float64, max_n_matches 20) on the headline
    list: the device TF-IDF matrix against sklearn and ALL 663 000 rows of the multiply against the port (VERDICT r04,
    missing 4 -- the every-row test above is fp32 / top 10)."""
    import os
    from string_grouper_amd.vectorizer import HipTfidfVectorizer
    n = 663000
    names = _names(n)
    vec = HipTfidfVectorizer(dtype=np.float64, ctx=ctx)
    prepared = vec.prepare(names)
    vec.fit_prepared([prepared])
    dA = vec.transform_prepared(prepared)
    A_ref = _tfidf(names, np.float64)
    assert_csr_identical(dA.to_scipy(), A_ref, "tf-idf at 663k, fp64")
    threads = max(1, min(64, len(os.sched_getaffinity(0))))
    C_ref = P.sp_matmul_topn_port(A_ref, A_ref.T, 20, 0.8, True, threads)
    post = ctx.postings_build(dA)
    res = ctx.spgemm_topn(dA, post, 20, 0.8, True)
    st = ctx.stats()
    assert st["prune_rows"] > 0 and st["prune_symmetric"] == 1 and st["exact_rows"] == 0, st
    assert_csr_identical(res.to_scipy(), C_ref, "663k self-join, fp64, top 20")
    for h in (res, post, dA):
        h.free()


@pytest.mark.timeout(900)
def test_config4_5M_selfjoin_fp64_more_than_100k_rows_equal_the_port(ctx):
    """configs[3] in the reference's default dtype: the whole 5 M self-join in fp64, the first / middle / last 30 000
    positions of the index (with all members of their groups) and the rows of special shape against the port."""
    from string_grouper_amd.vectorizer import HipTfidfVectorizer
    n = 5_000_000
    names = _names(n, seed=1234)
    vec = HipTfidfVectorizer(dtype=np.float64, ctx=ctx)
    p = vec.prepare(names)
    del names
    vec.fit_prepared([p])
    A = vec.transform_prepared(p)
    A_host = A.to_scipy()
    post = ctx.postings_build(A)
    res = ctx.spgemm_topn(A, post, 10, 0.8, True)
    st = ctx.stats()
    assert st["prune_symmetric"] == 1, st
    print(f"configs[3] fp64 whole: K4p group {st['ms_spgemm_topn']:.1f} ms, pairs scored {st['prune_survivors']:.3e}")
    rows = np.concatenate([_rows_at_position_blocks(ctx, post, n, 30_000), _rows_of_special_shape(A_host)])
    assert _rows_equal_the_port(res, rows, A_host, A_host, 10, 0.8, "configs[3], 5 M self-join, fp64") >= 100_000
    for h in (res, post, A):
        h.free()
    ctx.trim()


@pytest.mark.timeout(900)
def test_config5_asymmetric_10M_x_1M_fp64_more_than_100k_rows_equal_the_port(ctx):
    """configs[4] in the reference's default dtype: all 10 M master rows against 1 M duplicates, top 20 / 0.7, fp64; the
    first / middle / last 34 000 master rows and the rows of special shape against the port."""
    from string_grouper_amd.synth import synth_names
    from string_grouper_amd.vectorizer import HipTfidfVectorizer
    n_m, n_d = 10_000_000, 1_000_000
    master = _names(n_m, seed=1234)
    dupes = synth_names(n_d, seed=4321, perturb_of=master, perturb_frac=0.5)
    vec = HipTfidfVectorizer(dtype=np.float64, ctx=ctx)
    pm, pd_ = vec.prepare(master), vec.prepare(dupes)
    del master, dupes
    vec.fit_prepared([pm, pd_])
    A = vec.transform_prepared(pm)
    B = vec.transform_prepared(pd_)
    B_host = B.to_scipy()
    post = ctx.postings_build(B)
    res = ctx.spgemm_topn(A, post, 20, 0.7, True)
    st = ctx.stats()
    print(f"configs[4] fp64 whole: K4p group {st['ms_spgemm_topn']:.1f} ms, left rows multiplied {st['prune_rows']} of {n_m}")
    A_host = A.to_scipy()
    rows = np.concatenate([np.arange(34_000), n_m // 2 + np.arange(34_000), n_m - 34_000 + np.arange(34_000),
                           _rows_of_special_shape(A_host)])
    assert _rows_equal_the_port(res, rows, A_host, B_host, 20, 0.7, "configs[4], 10 M x 1 M, fp64") >= 100_000
    for h in (res, post, A, B):
        h.free()
    ctx.trim()


def test_second_filter_is_used_where_it_pays_and_changes_no_result(ctx):
    """Round 5: the 8-bit copies of the right-hand rows (sg_internal.h: SgScoreCtx::q8).  Same rows with the filter on, off
    (SG_Q8=0) and forced; `sg_stats` says what happened: at a name-matching threshold it rejects most candidates of the first
    filter; under 0.65 it is not used (every candidate is scored); for rows of more than 45 entries on average the records
    are not built unless forced."""
    names = _names(120000, seed=5)
    A = _tfidf(names, np.float32)
    want = {thr: P.sp_matmul_topn_port(A, A.T, 10, thr, True, 16) for thr in (0.8, 0.6)}
    seen = {}
    for q8 in (None, "0", "1"):
        ctx.set_option("SG_Q8", q8)
        dA = ctx.csr_from_scipy(A)
        post = ctx.postings_build(dA)
        for thr in (0.8, 0.6):
            res = ctx.spgemm_topn(dA, post, 10, thr, True)
            st = ctx.stats()
            assert_csr_identical(res.to_scipy(), want[thr], f"SG_Q8={q8} threshold {thr}")
            seen[(q8, thr)] = (st["prune_survivors"], st["prune_scored"], ctx.postings_bytes(post))
            res.free()
        post.free()
        dA.free()
    ctx.reset_options()
    for q8 in (None, "1"):
        cand, scored, _ = seen[(q8, 0.8)]
        assert cand > 100000 and scored * 10 < cand, seen          # most candidates never reach the exact scoring
        cand, scored, _ = seen[(q8, 0.6)]
        assert scored == cand, seen                                # under 0.65: filter not used, every candidate scored
    assert seen[("0", 0.8)][1] == seen[("0", 0.8)][0] and seen[("0", 0.8)][2] < seen[(None, 0.8)][2], seen   # no records built
    # rows of ~60 entries: no records unless forced; the result is the port's either way
    long_names = [" ".join(names[3 * i:3 * i + 3]) for i in range(20000)]
    L = _tfidf(long_names, np.float32)
    want_l = P.sp_matmul_topn_port(L, L.T, 10, 0.8, True, 16)
    sizes = {}
    for q8 in (None, "1"):
        ctx.set_option("SG_Q8", q8)
        dL = ctx.csr_from_scipy(L)
        post = ctx.postings_build(dL)
        res = ctx.spgemm_topn(dL, post, 10, 0.8, True)
        st = ctx.stats()
        assert_csr_identical(res.to_scipy(), want_l, f"long names, SG_Q8={q8}")
        sizes[q8] = (ctx.postings_bytes(post), st["prune_survivors"], st["prune_scored"])
        for h in (res, post, dL):
            h.free()
    ctx.reset_options()
    assert sizes[None][0] < sizes["1"][0] and sizes[None][1] == sizes[None][2] and sizes["1"][2] < sizes["1"][1], sizes


def test_bucket_walk_behind_a_full_bucket_keeps_every_hit(ctx):
    """Regression for the compiler finding of round 5 (scripts/q8_debug.py): hipcc 7.2 compiled the per-lane
    walk behind a FULL bucket of row i's hash with the hit read off the last trip's compare mask, and a row with five terms in
    one bucket lost four matches -- on exactly this list (20 000 synthetic names, seed 1234, every row indexed).  The walk is
    now a loop the wave leaves together; a toolchain bump that brings the miscompile back (in row_values or in any other
    divergent walk over the hash) fails here: second filter on == off == forced == the port, bit for bit, and the same on
    a list built to put MANY terms into one bucket."""
    names = _names(20000, seed=1234)
    A = _tfidf(names, np.float32)
    want = P.sp_matmul_topn_port(A, A.T, 10, 0.8, True, 16)
    for q8 in ("1", "0", None):
        ctx.set_option("SG_Q8", q8)
        ctx.set_option("SG_COLLAPSE", "0")
        dA = ctx.csr_from_scipy(A)
        post = ctx.postings_build(dA)
        res = ctx.spgemm_topn(dA, post, 10, 0.8, True)
        assert_csr_identical(res.to_scipy(), want, f"20 000 names, every row indexed, SG_Q8={q8}")
        for h in (res, post, dA):
            h.free()
    ctx.reset_options()
    # rows whose terms crowd a few buckets (the kernel's term_hash, restated: a bucket is four slots, five or more of a row's
    # terms in one bucket make lookups walk past it)
    rng = np.random.default_rng(7)
    n, V = 6000, 1 << 15
    bucket = (((np.arange(V, dtype=np.uint64) * np.uint64(0x9E3779)) & np.uint64(0xFFFFFFFF)) >> np.uint64(15)) & np.uint64(124)
    by_bucket = [np.flatnonzero(bucket == b) for b in range(0, 128, 4)]
    base_rows = []
    for _ in range(600):
        picks = [rng.choice(by_bucket[b], 7, replace=False) for b in rng.choice(32, 3, replace=False)]
        base_rows.append(np.unique(np.concatenate(picks + [rng.integers(0, V, 3)])))
    rows, cols, vals = [], [], []
    for i in range(n):
        k = base_rows[i % 600].copy()
        if i >= 600:                                   # near-duplicates: drop one term, add one
            k = np.unique(np.concatenate([np.delete(k, rng.integers(0, len(k))), [int(rng.integers(0, V))]]))
        v = rng.random(len(k)).astype(np.float32) + 0.5
        v /= np.float32(np.sqrt(np.sum(v.astype(np.float64) ** 2)) * 1.0000002)
        rows += [i] * len(k)
        cols += k.tolist()
        vals += v.tolist()
    M = sp.csr_matrix((np.asarray(vals, np.float32), (rows, cols)), shape=(n, V))
    M.sort_indices()
    want_m = P.sp_matmul_topn_port(M, M.T, 10, 0.6, True, 16)
    for q8 in ("1", "0"):
        ctx.set_option("SG_Q8", q8)
        ctx.set_option("SG_Q8_MIN_THRESHOLD", "0.5")
        dM = ctx.csr_from_scipy(M)
        post = ctx.postings_build(dM)
        res = ctx.spgemm_topn(dM, post, 10, 0.6, True)
        assert_csr_identical(res.to_scipy(), want_m, f"crowded buckets, SG_Q8={q8}")
        for h in (res, post, dM):
            h.free()
    ctx.reset_options()
