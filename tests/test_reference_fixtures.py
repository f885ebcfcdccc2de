"""The public API against fixtures made by the UNMODIFIED reference at 20-30 k names
(tests/golden/make_golden_large.py -> tests/golden/reference_large.npz): hubs of identical names, chains,
empty / short / non-ASCII rows, self-joins and master x duplicates, float32 and float64, both group_rep values.

* CPU (``-m "not gpu"``): the host mirror on top of the oracle engine must reproduce every fixture -- pins the
  mirror's host logic (frames, symmetrisation, groups, nearest matches) on the reference itself at scale.
* GPU (``-m gpu``): the same through the HIP engine -- K1-K4p/K4, the fused tail K6 and the reductions K7 / K8
  against the reference's output (string_grouper.py:417-431, :783-849, :851-904), not against the mirror.
Indices are compared exactly, similarities bit for bit (float64)."""
import os

import numpy as np
import pandas as pd
import pytest

from tests import _fixture_inputs as F

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = np.load(os.path.join(HERE, "golden", "reference_large.npz"))


def run_case(api, name):
    kind, spec, kw = F.CASES[name]
    master, dups = F.build_inputs(spec)
    kwargs = F.resolve_kwargs(kw)
    m = pd.Series(master, name="name")
    d = None if dups is None else pd.Series(dups, name="dup")
    if kind == "match_strings":
        df = api.match_strings(m, d, **kwargs)
        np.testing.assert_array_equal(df["left_index"].to_numpy(dtype=np.int64), GOLD[name + "/left_index"], err_msg=name)
        np.testing.assert_array_equal(df["right_index"].to_numpy(dtype=np.int64), GOLD[name + "/right_index"], err_msg=name)
        np.testing.assert_array_equal(df["similarity"].to_numpy(dtype=np.float64), GOLD[name + "/similarity"], err_msg=name)
        assert (df["left_name"].to_numpy() == m.to_numpy()[GOLD[name + "/left_index"]]).all()
    elif kind == "group_similar_strings":
        g = api.group_similar_strings(m, **kwargs)
        want = GOLD[name + "/group_rep_index"]
        np.testing.assert_array_equal(g["group_rep_index"].to_numpy(dtype=np.int64), want, err_msg=name)
        assert (g["group_rep_name"].to_numpy() == m.to_numpy()[want]).all()
    else:
        r = api.match_most_similar(m, d, **kwargs)
        want = GOLD[name + "/most_similar_index"]
        idx = r["most_similar_index"].to_numpy()
        matched = ~pd.isna(idx)
        np.testing.assert_array_equal(matched, want >= 0, err_msg=name)
        np.testing.assert_array_equal(idx[matched].astype(np.int64), want[matched], err_msg=name)
        strs = r["most_similar_name"].to_numpy()
        assert (strs[matched] == m.to_numpy()[want[matched]]).all() and (strs[~matched] == d.to_numpy()[~matched]).all()


@pytest.fixture()
def oracle_api():
    import string_grouper_amd as sga
    import string_grouper_amd.engine as E
    from tests._oracle_engine import OracleEngine
    old = E._engine
    E.set_engine(OracleEngine(use_port=True))
    yield sga
    E.set_engine(old)


@pytest.fixture()
def hip_api(ctx):
    import string_grouper_amd as sga
    import string_grouper_amd.engine as E
    old = E._engine
    E.set_engine(E.HipEngine(ctx))
    yield sga
    E.set_engine(old)


@pytest.mark.parametrize("name", sorted(F.CASES))
def test_host_mirror_reproduces_reference_fixture(oracle_api, name):
    run_case(oracle_api, name)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(F.CASES))
def test_hip_engine_reproduces_reference_fixture(hip_api, name):
    run_case(hip_api, name)
