"""CPU tests of the host-side mirror (string_grouper_amd/string_grouper.py) with an engine double:
the reference's public behaviour, its golden vectors, and -- when the reference tree is mounted --
the reference's own unit tests executed against the mirror."""
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np
import pandas as pd
import pytest

from tests import _golden as G
from tests._oracle_engine import OracleEngine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture()
def api():
    import string_grouper_amd
    import string_grouper_amd.engine as E
    old = E._engine
    eng = OracleEngine()
    E.set_engine(eng)
    string_grouper_amd._test_engine = eng
    yield string_grouper_amd
    E.set_engine(old)


def test_golden_cases_through_public_api(api):
    G.run_api_checks(api)


def test_product_engine_refuses_to_run_without_gpu():
    """No CPU fallback: with no engine injected the package needs the HIP library AND a GPU."""
    import string_grouper_amd.engine as E
    from string_grouper_amd import _native as N
    old = E._engine
    E.set_engine(None)
    try:
        if N.device_count() == 0:
            with pytest.raises(Exception) as ei:
                E.get_engine().ctx
            assert "no HIP device" in str(ei.value) or "gfx950" in str(ei.value)
    finally:
        E.set_engine(old)


def test_config_and_validation(api):
    cfg = api.StringGrouperConfig()
    assert (cfg.ngram_size, cfg.max_n_matches, cfg.min_similarity, cfg.regex) == (3, 20, 0.8, r'[,-./]|\s')
    assert cfg.tfidf_matrix_dtype is np.float64 and cfg.n_blocks is None and cfg.normalize_to_ascii
    with pytest.raises(Exception):
        cfg.min_similarity = 0.1
    with pytest.raises(TypeError):
        api.StringGrouper(pd.Series(["a"]), not_an_option=1)
    s = pd.Series(["foo", "bar"])
    for bad in (2, (0, 2), (1, 2.5), (1, 2, 3), (1,)):
        with pytest.raises(Exception):
            api.match_strings(s, n_blocks=bad)
    for bad in (None, 0, "whatever"):
        with pytest.raises(Exception):
            api.match_strings(s, tfidf_matrix_dtype=bad)
    with pytest.raises(TypeError):
        api.StringGrouper("foo", "bar")
    with pytest.raises(TypeError):
        api.StringGrouper(pd.Series(["foo", "bar"]), pd.Series(["foo", 1]))
    with pytest.raises(TypeError):
        api.StringGrouper(pd.Series(["foo", np.nan]))
    with pytest.raises(api.StringGrouperNotFitException):
        api.StringGrouper(s).get_matches()
    with pytest.raises(Exception):
        api.StringGrouper(s, duplicates_id=pd.Series([1, 2]))
    with pytest.raises(Exception):
        api.group_similar_strings(s, group_rep="nonsense")


def test_guessed_blocks_use_one_device_multiply_and_explicit_blocks_are_honoured(api):
    eng = api._test_engine
    names = pd.Series(G.INPUTS["accounts_names"])
    ref = api.match_strings(names, min_similarity=0.5)
    assert [c[0] for c in eng.calls] == ["single"]
    eng.calls.clear()
    got = api.match_strings(names, min_similarity=0.5, n_blocks=(2, 3))
    assert eng.calls == [("blocked", (14, 74), (14, 74), (2, 3))]
    key = ["left_index", "right_index"]
    pd.testing.assert_frame_equal(ref.sort_values(key).reset_index(drop=True), got.sort_values(key).reset_index(drop=True))
    assert got.similarity.dtype == np.float64


def test_overflow_error_triggers_the_reference_fallback(api):
    """fit() retries block-wise when the single multiply reports OverflowError
    (string_grouper.py:397-413); SG_ERR_OVERFLOW of the C ABI maps to that exception."""
    names = pd.Series(G.INPUTS["customers2"])
    sg = api.StringGrouper(names, min_similarity=0.1)
    real = sg._build_matches
    seen = []

    def flaky(a, b, n_blocks):
        seen.append(n_blocks)
        if len(seen) == 1:
            raise OverflowError
        return real(a, b, n_blocks)
    sg._build_matches = flaky
    ref = api.match_strings(names, min_similarity=0.1)
    got = sg.match_strings(names, n_blocks=(1, 1))
    assert len(seen) == 2 and seen[0] == (1, 1)
    pd.testing.assert_frame_equal(ref, got)


def test_add_and_remove_match(api):
    s = pd.Series(['foooo', 'no match', 'baz', 'foooo'])
    sg = api.StringGrouper(s).fit()
    sg.add_match('no match', 'baz')
    m = sg.get_matches()
    assert len(m[(m.left_side == 'no match') & (m.right_side == 'baz')]) == 1
    assert len(m[(m.left_side == 'baz') & (m.right_side == 'no match')]) == 1
    with pytest.raises(ValueError):
        sg.add_match('doesnt exist', 'baz')
    sg2 = api.StringGrouper(pd.Series(['foooo', 'no match', 'baz', 'foooob'])).fit()
    sg2.remove_match('foooo', 'foooob')
    m = sg2.get_matches()
    assert len(m[(m.left_side == 'foooo') & (m.right_side == 'foooob')]) == 0
    assert len(m[(m.left_side == 'foooob') & (m.right_side == 'foooo')]) == 0


def test_drop_in_alias_is_the_same_module(api):
    import string_grouper
    import string_grouper.string_grouper as inner
    import string_grouper_amd.string_grouper as impl
    assert inner is impl and string_grouper.match_strings is impl.match_strings


@pytest.mark.skipif(not os.path.isdir("/root/reference/string_grouper"), reason="reference tree not mounted")
def test_reference_unit_tests_pass_against_the_mirror():
    """The reference's 53 unit tests, unmodified (copied to a temp dir only for collection), run
    against this package through the drop-in alias with the oracle engine injected."""
    d = tempfile.mkdtemp()
    try:
        shutil.copy("/root/reference/string_grouper/test/test_string_grouper.py", os.path.join(d, "test_ref_copy.py"))
        with open(os.path.join(d, "conftest.py"), "w") as f:
            f.write("import sys\nsys.path.insert(0, %r)\nsys.path.insert(0, %r)\n"
                    "from _oracle_engine import OracleEngine\nimport string_grouper_amd.engine as E\n"
                    "E.set_engine(OracleEngine())\nimport string_grouper\n" % (ROOT, os.path.join(ROOT, "tests")))
        r = subprocess.run([sys.executable, "-m", "pytest", d, "-q", "-p", "no:cacheprovider", "--rootdir=" + d],
                           cwd=d, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "53 passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
    finally:
        shutil.rmtree(d, ignore_errors=True)


def test_assigning_the_match_list_drops_a_device_copy():
    """K7/K8 reduce the device-resident match list of the last fit(); any host-side edit of the list
    (add_match, remove_match, a test assigning the attribute) must invalidate that copy."""
    import pandas as pd
    import string_grouper_amd as sga

    class FakeDeviceList:
        freed = False

        def free(self):
            self.freed = True

    sg = sga.StringGrouper(pd.Series(["a b c", "a b d"]))
    fake = FakeDeviceList()
    sg.__dict__["_device_matches"] = fake
    sg._matches_list = pd.DataFrame({"master_side": [0], "dupe_side": [0], "similarity": [1.0]})
    assert fake.freed and "_device_matches" not in sg.__dict__
    assert len(sg._matches_list) == 1


def test_pruned_multiply_rule_used_for_the_row_split():
    from string_grouper_amd.distributed import pruned_multiply_expected
    assert pruned_multiply_expected(10, 0.8)
    assert pruned_multiply_expected(128, 0.8)          # (65 .. 128: the pruned kernel + a hand-over of full rows)
    assert not pruned_multiply_expected(129, 0.8)
    assert not pruned_multiply_expected(10, 0.3)
    # round 6: the tile-by-tile form's envelope starts at 0.40 (sg_spgemm_topn.hip, prune_min_threshold) -- 0.45 with the form off

    class _Ctx:
        def __init__(self, opts):
            self._o = opts

        def options(self):
            return self._o
    assert pruned_multiply_expected(10, 0.42, _Ctx({})) and not pruned_multiply_expected(10, 0.39, _Ctx({}))
    assert not pruned_multiply_expected(10, 0.42, _Ctx({"SG_ALT_FORM": "0"}))
    assert pruned_multiply_expected(10, 0.2, _Ctx({"SG_PRUNE_MIN_THRESHOLD": "0.1"})) and not pruned_multiply_expected(10, 0.9, _Ctx({"SG_PRUNE": "0"}))


# ------------------------------------------------------------------------------------------------
# The glue around the device reductions (K7 / K8 / K9), exercised on the CPU with an engine double whose
# "device" list reduces with numpy: the frames built from the reduced arrays must equal the ones the
# host formulation builds from the full match list.
class _NumpyDeviceList:
    def __init__(self, rows, cols, vals, n_cols):
        self.rows, self.cols, self.vals, self.n_cols = rows, cols, vals, n_cols
        self.freed = False

    def best_master(self):
        import numpy as np
        best = np.full(self.n_cols, -1, np.int32)
        order = np.lexsort((self.rows, -self.vals.astype(np.float64), self.cols))
        cs = self.cols[order]
        first = np.ones(len(order), bool)
        first[1:] = cs[1:] != cs[:-1]
        best[cs[first]] = self.rows[order][first]
        return best

    def group_reps(self, centroid):
        import numpy as np
        import scipy.sparse as sp
        from scipy.sparse.csgraph import connected_components
        n = self.n_cols
        g = sp.csr_matrix((np.ones(len(self.rows)), (self.rows, self.cols)), shape=(n, n))
        _, labels = connected_components(g, directed=True)
        if centroid:
            g.data = self.vals.astype(np.float64)
            weight = np.asarray(g.sum(axis=1)).squeeze(axis=1)
            order = np.lexsort((np.arange(n), -weight, labels))
        else:
            order = np.lexsort((np.arange(n), labels))
        ls = labels[order]
        head = np.ones(n, bool)
        head[1:] = ls[1:] != ls[:-1]
        rep_of = np.empty(labels.max() + 1, np.int64)
        rep_of[ls[head]] = order[head]
        return rep_of[labels].astype(np.int32)

    def free(self):
        self.freed = True


def _engine_with_device_reductions():
    import numpy as np
    import scipy.sparse as sp
    from tests._oracle_engine import OracleEngine

    class Engine(OracleEngine):
        def match_list(self, A, B, top_n, threshold, self_join_fix, keep_on_device=False):
            import string_grouper_amd as sga
            C = self._mul(A.m, B.m, top_n, threshold)
            true_max = int(np.diff(C.indptr).max()) if C.shape[0] else 0
            if self_join_fix:
                C = sga.StringGrouper._symmetrize_matrix(sga.StringGrouper._fix_diagonal(C))
            C = sp.csr_matrix(C)
            rows = np.repeat(np.arange(C.shape[0], dtype=np.int64), np.diff(C.indptr))
            cols, vals = C.indices.astype(np.int64), C.data
            out = (rows, cols, vals, true_max)
            return out + (_NumpyDeviceList(rows, cols, vals, B.shape[0]),) if keep_on_device else out

        def rowwise_dot(self, A, B):
            return np.asarray(A.m.multiply(B.m).sum(axis=1)).squeeze(axis=1)

    return Engine(use_port=True)


def test_get_groups_through_the_device_reductions_equals_the_host_formulation():
    import numpy as np
    import pandas as pd
    import string_grouper_amd as sga
    import string_grouper_amd.engine as E
    from string_grouper_amd.synth import synth_names
    old = E._engine
    E.set_engine(_engine_with_device_reductions())
    try:
        names = list(synth_names(1500, 5))
        s = pd.Series(names + [names[3]] * 7, index=np.arange(1507) * 3)
        for rep in ("centroid", "first"):
            sg = sga.StringGrouper(s, min_similarity=0.8, group_rep=rep).fit()
            dml = sg.__dict__["_device_matches"]
            on_device = sg.get_groups()
            sg._drop_device_matches()
            assert dml.freed
            pd.testing.assert_frame_equal(on_device, sg.get_groups())
        dupes = pd.Series(synth_names(700, seed=6, perturb_of=names, perturb_frac=0.7))
        ids_m, ids_d = pd.Series(np.arange(len(s)) + 100), pd.Series(np.arange(len(dupes)) + 9000)
        for kw in (dict(), dict(ignore_index=True)):
            sg = sga.StringGrouper(s.reset_index(drop=True), dupes, ids_m, ids_d, min_similarity=0.7, **kw).fit()
            on_device = sg.get_groups()
            sg._drop_device_matches()
            pd.testing.assert_frame_equal(on_device, sg.get_groups())
        sg = sga.StringGrouper(s.reset_index(drop=True), dupes, min_similarity=0.7, replace_na=True).fit()
        on_device = sg.get_groups()
        sg._drop_device_matches()
        pd.testing.assert_frame_equal(on_device, sg.get_groups())
        sims = sga.compute_pairwise_similarities(pd.Series(names[:700]), dupes)
        from oracle import oracle as O
        (a, b), _, _ = O.tfidf_sklearn(names[:700] + list(dupes), [names[:700], list(dupes)], dtype=np.float64)
        assert sims.name == "similarity" and len(sims) == 700
        np.testing.assert_array_equal(sims.to_numpy(), np.asarray(a.multiply(b).sum(axis=1)).squeeze(axis=1))
    finally:
        E.set_engine(old)


@pytest.mark.skipif(not os.path.isdir("/root/reference/string_grouper"), reason="reference tree not mounted")
def test_public_and_private_surface_matches_the_reference():
    """Every name of the reference module that a user (or the reference's tests) can reach exists here with the same
    parameters and defaults: module constants, the four top-level functions, StringGrouperConfig, and every method and
    attribute of StringGrouper incl. the private ones (string_grouper/string_grouper.py)."""
    import importlib
    import inspect
    code = r"""
import sys, inspect, importlib, json
sys.path[:0] = [%r, "/root/reference"]
ref = importlib.import_module("string_grouper.string_grouper")
out = {"consts": {n: repr(getattr(ref, n)) for n in dir(ref) if n.isupper()},
       "funcs": {n: [(p.name, repr(p.default), str(p.kind)) for p in inspect.signature(o).parameters.values()]
                 for n, o in vars(ref).items() if inspect.isfunction(o) and o.__module__ == ref.__name__},
       "methods": {n: ([(p.name, repr(p.default), str(p.kind)) for p in inspect.signature(o).parameters.values()]
                       if callable(o) else None)
                   for n, o in inspect.getmembers(ref.StringGrouper) if not (n.startswith("__") and n != "__init__")},
       "config": [list(ref.StringGrouperConfig._fields), {k: repr(v) for k, v in ref.StringGrouperConfig._field_defaults.items()}]}
print(json.dumps(out))
""" % os.path.join(ROOT, "tests", "ref_shims")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd="/tmp")
    assert r.returncode == 0, r.stderr[-2000:]
    import json
    want = json.loads(r.stdout.strip().splitlines()[-1])
    import string_grouper_amd.string_grouper as mine

    def params(o):
        return [[p.name, repr(p.default), str(p.kind)] for p in inspect.signature(o).parameters.values()]

    for n, v in want["consts"].items():
        assert hasattr(mine, n) and repr(getattr(mine, n)) == v, n
    for n, sig in want["funcs"].items():
        if n == "validate_is_fit":                      # a decorator: its one parameter's name is nobody's interface
            assert hasattr(mine, n)
            continue
        assert hasattr(mine, n), n
        assert params(getattr(mine, n)) == sig, (n, params(getattr(mine, n)), sig)
    for n, sig in want["methods"].items():
        assert hasattr(mine.StringGrouper, n), n
        if sig is not None:
            assert params(getattr(mine.StringGrouper, n)) == sig, (n, params(getattr(mine.StringGrouper, n)), sig)
    assert [list(mine.StringGrouperConfig._fields), {k: repr(v) for k, v in mine.StringGrouperConfig._field_defaults.items()}] \
        == want["config"]


_INDEX_CASES = """
import numpy as np, pandas as pd
base = ["ACME HOLDINGS INC", "ACME HOLDING INC", "Acme Holdings, Inc.", "ZENITH PARTNERS LP", "ZENITH PARTNER LP",
        "OMEGA CAPITAL LLC", "OMEGA CAPITAL, LLC", "BOREAL TRUST", "BOREAL TRUST CO", "QUARTZ FUND", "QUARTZ FUNDS",
        "LONE STRING WITHOUT A MATCH"]
dup = ["ACME HOLDINGS", "ZENITH PARTNERS", "OMEGA CAPITAL", "NOTHING LIKE THE OTHERS AT ALL", "QUARTZ FUND LP"]


def cases():
    n, m = len(base), len(dup)
    yield "default index", pd.Series(base), None, None, None, {}
    yield "named series, range index with start and step", pd.Series(base, index=pd.RangeIndex(10, 10 + 3 * n, 3), name="name"), None, None, None, {}
    yield "string index with a name", pd.Series(base, index=pd.Index([f"k{i:02d}" for i in range(n)], name="key")), None, None, None, {}
    yield "unsorted integer index", pd.Series(base, index=np.arange(n)[::-1] * 7 + 1), None, None, None, {}
    yield "datetime index", pd.Series(base, index=pd.date_range("2020-01-01", periods=n, freq="D")), None, None, None, {}
    yield "float index", pd.Series(base, index=np.linspace(0.5, 6.0, n)), None, None, None, {}
    yield "multi index", pd.Series(base, index=pd.MultiIndex.from_arrays([np.arange(n) // 3, np.arange(n) % 3], names=["a", "b"])), None, None, None, {}
    yield "series named index", pd.Series(base, name="index"), None, None, None, {}
    yield "ignore_index", pd.Series(base, index=[f"k{i}" for i in range(n)]), None, None, None, {"ignore_index": True}
    yield "with ids", pd.Series(base, name="nm"), None, pd.Series(np.arange(n) + 100, name="uid"), None, {}
    yield "master x duplicates", pd.Series(base, index=pd.RangeIndex(5, 5 + n)), pd.Series(dup, index=[f"d{i}" for i in range(m)], name="dupe"), None, None, {}
    yield "master x duplicates with ids", pd.Series(base), pd.Series(dup), pd.Series(np.arange(n) * 2, name="mid"), pd.Series([f"x{i}" for i in range(m)], name="did"), {}
    yield "string dtype", pd.Series(base, dtype="string", name="s"), None, None, None, {}
    yield "arrow string dtype", pd.Series(base, dtype="string[pyarrow]", index=[f"r{i}" for i in range(n)]), pd.Series(dup, dtype="string[pyarrow]"), None, None, {}
    yield "zeroes included", pd.Series(base[:6], index=pd.RangeIndex(3, 9)), None, None, None, {"min_similarity": 0.0, "include_zeroes": True, "max_n_matches": 6}
"""


@pytest.mark.skipif(not os.path.isdir("/root/reference/string_grouper"), reason="reference tree not mounted")
def test_frames_equal_the_references_for_every_kind_of_index():
    """get_matches() / get_groups() assemble their frames without pandas' take + reset_index on the common shapes
    (string_grouper_amd/string_grouper.py: side()).  Whatever the index of the input Series is -- default, a range with
    start and step, strings, unsorted integers, datetimes, floats, a MultiIndex, names that collide with reset_index's --
    the frames must be the unmodified reference's: values, dtypes, column names and order."""
    import pickle
    d = tempfile.mkdtemp()
    try:
        with open(os.path.join(d, "cases.py"), "w") as f:
            f.write(_INDEX_CASES)
        ref_code = (
            "import sys, pickle\n"
            "sys.path[:0] = [%r, %r, '/root/reference']\n"
            "import os\nos.environ['SG_SHIM_BACKEND'] = 'oracle'\n"
            "from cases import cases\n"
            "from string_grouper import match_strings, group_similar_strings, match_most_similar\n"
            "out = {}\n"
            "for name, m, dd, mid, did, kw in cases():\n"
            "    out[name, 'match'] = match_strings(m, dd, mid, did, **kw)\n"
            "    if dd is None and 'include_zeroes' not in kw:\n"
            "        out[name, 'groups'] = group_similar_strings(m, mid, **kw)\n"
            "    if dd is not None:\n"
            "        out[name, 'nearest'] = match_most_similar(m, dd, mid, did, **kw)\n"
            "pickle.dump(out, open(%r, 'wb'))\n"
        ) % (d, os.path.join(ROOT, "tests", "ref_shims"), os.path.join(d, "ref.pkl"))
        r = subprocess.run([sys.executable, "-c", ref_code], cwd=d, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        want = pickle.load(open(os.path.join(d, "ref.pkl"), "rb"))
        sys.path.insert(0, d)
        try:
            import importlib
            cases_mod = importlib.import_module("cases")
        finally:
            sys.path.remove(d)
        import string_grouper_amd as sga
        import string_grouper_amd.engine as E
        old = E._engine
        E.set_engine(OracleEngine())
        try:
            for name, m, dd, mid, did, kw in cases_mod.cases():
                got = sga.match_strings(m, dd, mid, did, **kw)
                pd.testing.assert_frame_equal(got, want[name, "match"], obj=f"match_strings, {name}")
                if dd is None and "include_zeroes" not in kw:
                    g = sga.group_similar_strings(m, mid, **kw)
                    w = want[name, "groups"]
                    (pd.testing.assert_frame_equal if isinstance(w, pd.DataFrame) else pd.testing.assert_series_equal)(
                        g, w, obj=f"group_similar_strings, {name}")
                if dd is not None:
                    g = sga.match_most_similar(m, dd, mid, did, **kw)
                    w = want[name, "nearest"]
                    (pd.testing.assert_frame_equal if isinstance(w, pd.DataFrame) else pd.testing.assert_series_equal)(
                        g, w, obj=f"match_most_similar, {name}")
        finally:
            E.set_engine(old)
            sys.modules.pop("cases", None)
    finally:
        shutil.rmtree(d, ignore_errors=True)




@pytest.mark.skipif(not os.path.isdir("/root/reference/string_grouper"), reason="reference tree not mounted")
def test_seeded_fuzz_of_the_public_api_against_the_reference():
    """150 random small jobs -- options drawn at random (thresholds, max_n_matches, n-gram sizes, regexes incl. ones that
    are not character classes, case / ASCII handling on strings with accents, sharp s, Greek final sigma, dtypes,
    n_blocks, ids, group representatives, replace_na) -- through all five entry points: the frames / Series (or the
    exception) must be the unmodified reference's."""
    import pickle
    d = tempfile.mkdtemp()
    try:
        ref_code = (
            "import sys, pickle, os\n"
            "sys.path[:0] = [%r, %r, '/root/reference']\n"
            "sys.path.append(%r)\n"
            "os.environ['SG_SHIM_BACKEND'] = 'oracle'\n"
            "import string_grouper as api\n"
            "assert api.__file__.startswith('/root/reference'), api.__file__\n"
            "from tests._fuzz_cases import cases, run\n"
            "out = {c: run(api, kind, m, dd, mid, did, kw) for c, kind, m, dd, mid, did, kw in cases()}\n"
            "pickle.dump(out, open(%r, 'wb'))\n"
        ) % (d, os.path.join(ROOT, "tests", "ref_shims"), ROOT, os.path.join(d, "ref.pkl"))
        r = subprocess.run([sys.executable, "-c", ref_code], cwd=d, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        want = pickle.load(open(os.path.join(d, "ref.pkl"), "rb"))
        from tests import _fuzz_cases as fz
        import string_grouper_amd as sga
        import string_grouper_amd.engine as E
        old = E._engine
        E.set_engine(OracleEngine())
        try:
            n_raised = 0
            for c, kind, m, dd, mid, did, kw in fz.cases():
                got, w = fz.run(sga, kind, m, dd, mid, did, kw), want[c]
                what = f"case {c}: {kind} {kw}"
                if isinstance(w, tuple):
                    n_raised += 1
                    assert isinstance(got, tuple) and got[:2] == w[:2], (what, got, w)
                elif isinstance(w, pd.DataFrame):
                    pd.testing.assert_frame_equal(got, w, obj=what)
                else:
                    pd.testing.assert_series_equal(got, w, obj=what)
            assert n_raised < 50
        finally:
            E.set_engine(old)
    finally:
        shutil.rmtree(d, ignore_errors=True)



@pytest.mark.skipif(not os.path.isdir("/root/reference/string_grouper"), reason="reference tree not mounted")
def test_series_validation_answers_as_the_reference_does():
    """``_is_series_of_strings`` (string_grouper.py:988-995 of the reference) on every kind of Series a caller may pass:
    object / string / arrow / categorical columns, empty ones of any dtype, missing values, mixed content."""
    code = r"""
import sys, json
sys.path[:0] = [%r, "/root/reference"]
import numpy as np, pandas as pd
from string_grouper.string_grouper import StringGrouper as R
sys.path.insert(0, %r)
from string_grouper_amd.string_grouper import StringGrouper as M
cases = {
    "object str": pd.Series(["a", "b"]), "string dtype": pd.Series(["a", "b"], dtype="string"),
    "arrow": pd.Series(["a", "b"], dtype="string[pyarrow]"), "categorical of str": pd.Series(["a", "b", "a"], dtype="category"),
    "categorical of int": pd.Series(pd.Categorical([1, 2])), "mixed": pd.Series(["a", 1]), "none": pd.Series(["a", None]),
    "nan": pd.Series(["a", np.nan]), "pd.NA in string dtype": pd.Series(["a", pd.NA], dtype="string"),
    "ints": pd.Series([1, 2]), "floats": pd.Series([1.0]), "bytes": pd.Series([b"a"]),
    "empty object": pd.Series([], dtype=object), "empty float": pd.Series([], dtype=float),
    "empty string": pd.Series([], dtype="string"), "empty category": pd.Series([], dtype="category"),
    "not a series": ["a", "b"], "frame": pd.DataFrame({"a": ["x"]}), "index": pd.Index(["a"]),
    "datetimes": pd.Series(pd.date_range("2020", periods=2)), "bools": pd.Series([True, False]),
}
def ask(f, v):
    try:
        return bool(f(v))
    except Exception as e:           # (the reference's map/any chain raises on some extension dtypes: nothing to compare)
        return "raises " + type(e).__name__
print(json.dumps({k: [ask(R._is_series_of_strings, v), ask(M._is_series_of_strings, v)] for k, v in cases.items()}))
""" % (os.path.join(ROOT, "tests", "ref_shims"), ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd="/tmp")
    assert r.returncode == 0, r.stderr[-2000:]
    import json
    got = json.loads(r.stdout.strip().splitlines()[-1])
    assert len(got) > 15
    compared = 0
    for k, (ref, mine) in got.items():
        assert isinstance(mine, bool), (k, mine)                 # the mirror never raises on a Series
        if isinstance(ref, bool):
            assert ref == mine, (k, ref, mine)
            compared += 1
    assert compared > 15


# ------------------------------------------------------------------------------------------------
# Seam b1 by name (round 4): the constructor call the reference makes, served by the device vectoriser's front end
def test_tfidf_vectorizer_by_name_takes_its_options_from_the_bound_analyzer():
    import pandas as pd
    import string_grouper_amd as sga
    from string_grouper_amd.vectorizer import TfidfVectorizer
    sg = sga.StringGrouper(pd.Series(["Acme Inc", "Ácme Corp"]), ngram_size=4, regex=r"[ie]", ignore_case=False,
                           normalize_to_ascii=False, tfidf_matrix_dtype=np.float32)
    vec = TfidfVectorizer(min_df=1, analyzer=sg.n_grams, dtype=np.float32)       # string_grouper.py:306, verbatim
    assert (vec.ngram_size, vec.regex, vec.ignore_case, vec.normalize_to_ascii, vec.dtype) == (4, r"[ie]", False, False, np.float32)
    with pytest.raises(TypeError):
        TfidfVectorizer(min_df=1, analyzer=lambda s: list(s), dtype=np.float32)      # code the device cannot run
    with pytest.raises(TypeError):
        TfidfVectorizer(min_df=1, analyzer=sg.n_grams, dtype=np.float32, sublinear_tf=True)
    with pytest.raises(NotImplementedError):
        TfidfVectorizer(min_df=2, analyzer=sg.n_grams, dtype=np.float32)


@pytest.mark.skipif(not os.path.isdir("/root/reference/string_grouper"), reason="reference tree not mounted")
def test_reference_package_is_packed_for_the_gpu_box():
    """oracle/mount_reference.py: the archive the GPU suite runs the unmodified reference from (a build output, git-ignored)."""
    import subprocess
    import zipfile
    from oracle import mount_reference
    path = mount_reference.mount()
    assert path and os.path.exists(path)
    names = zipfile.ZipFile(path).namelist()
    assert "string_grouper/string_grouper.py" in names and "string_grouper/test/test_string_grouper.py" in names
    tracked = subprocess.run(["git", "ls-files", "oracle/_ref"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
    assert tracked == "", "nothing under oracle/_ref may enter the history"
