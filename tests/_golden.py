"""Loader + checkers for tests/golden/reference_golden.json (made by tests/golden/make_golden.py)."""
import json
import os

import numpy as np
import pandas as pd
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "reference_golden.json")) as f:
    GOLD = json.load(f)
INPUTS = GOLD["inputs"]
CASES = GOLD["cases"]
KNOWN = GOLD["known_answers"]


def csr_from_golden(g):
    data = np.array([float.fromhex(x) for x in g["data_hex"]], dtype=np.float64).astype(g["dtype"])
    return sp.csr_matrix((data, np.array(g["indices"], np.int32), np.array(g["indptr"], np.int32)), shape=tuple(g["shape"]))


def frame_from_golden(g):
    return pd.DataFrame(g["rows"], columns=g["columns"])


def kwargs_from_golden(kw):
    kw = dict(kw)
    if "tfidf_matrix_dtype" in kw:
        kw["tfidf_matrix_dtype"] = getattr(np, kw["tfidf_matrix_dtype"])
    if "n_blocks" in kw:
        kw["n_blocks"] = tuple(kw["n_blocks"])
    return kw


def assert_csr_bitequal(a, b, what=""):
    a = sp.csr_matrix(a).copy()
    b = sp.csr_matrix(b).copy()
    a.sort_indices()
    b.sort_indices()
    assert a.shape == b.shape, what
    np.testing.assert_array_equal(a.indptr, b.indptr, err_msg=what)
    np.testing.assert_array_equal(a.indices, b.indices, err_msg=what)
    assert a.dtype == b.dtype, what
    np.testing.assert_array_equal(a.data, b.data, err_msg=what)


def assert_frame_matches_golden(df, g, what=""):
    """Indices / strings exact, similarity bit-exact (the golden float went through JSON repr, which
    round-trips doubles exactly)."""
    exp = frame_from_golden(g)
    got = df.reset_index(drop=True)
    assert list(map(str, got.columns)) == list(exp.columns), what
    for c in exp.columns:
        if c == "similarity":
            np.testing.assert_array_equal(got[c].to_numpy(dtype=np.float64), exp[c].to_numpy(dtype=np.float64), err_msg=what)
        else:
            pd.testing.assert_series_equal(got[c], exp[c], check_dtype=False, check_names=False, obj=f"{what}: column {c}")


def run_api_checks(api):
    """``api``: a module-like object with the reference's public names.  Runs every golden case that
    goes through the public API.  Used with the oracle engine (CPU) and the HIP engine (GPU)."""
    acc = pd.Series(INPUTS["accounts_names"], name="name")
    acc_ids = pd.Series(INPUTS["accounts_ids"], name="id")
    cust = pd.Series(INPUTS["customers"], name="Customer Name")
    cust2 = pd.Series(INPUTS["customers2"], name="Customer Name")
    for name in ("accounts_default", "accounts_min07", "accounts_top2", "accounts_f32", "accounts_blocks_2_3"):
        case = CASES["match_strings_" + name]
        df = api.match_strings(acc, **kwargs_from_golden(case["kwargs"]))
        assert_frame_matches_golden(df, case["frame"], name)
    df = api.match_strings(cust, cust2, min_similarity=0.1)
    assert_frame_matches_golden(df, CASES["match_strings_customers_vs_customers2_min01"]["frame"], "cust vs cust2")
    df = api.match_strings(acc, master_id=acc_ids, min_similarity=0.7)
    assert_frame_matches_golden(df, CASES["match_strings_accounts_with_ids_min07"]["frame"], "ids")
    g = api.group_similar_strings(acc, min_similarity=0.7)
    assert_frame_matches_golden(g, CASES["group_accounts_min07"]["frame"], "groups")
    g = api.group_similar_strings(acc, min_similarity=0.7, group_rep="first", ignore_index=True)
    assert g.tolist() == CASES["group_accounts_min07_first"]["values"]
    m = api.match_most_similar(cust, pd.Series(INPUTS["customers2"] + ["nothing alike"], name="dup"), min_similarity=0.6)
    assert_frame_matches_golden(m, CASES["most_similar_customers"]["frame"], "most similar")
    # known answers of the reference's own tests
    ka = KNOWN
    sg = api.StringGrouper(pd.Series(["aaa"]), ignore_case=False)
    assert sg.n_grams("McDonalds") == ka["ngrams_McDonalds_case"]["value"]
    sg = api.StringGrouper(pd.Series(["aaa"]))
    assert sg.n_grams("McDonalds") == ka["ngrams_McDonalds_lower"]["value"]
    assert sg.n_grams(ka["ngrams_unicode"]["input"]) == ka["ngrams_unicode"]["value"]
    sg = api.StringGrouper(pd.Series(ka["tfidf_foo_bar_baz"]["input"]))
    a, b = sg._get_tf_idf_matrices()
    np.testing.assert_array_equal(a.toarray(), np.array(ka["tfidf_foo_bar_baz"]["dense"]))
    sg = api.StringGrouper(pd.Series(ka["tfidf_master_dupes"]["master"]), pd.Series(ka["tfidf_master_dupes"]["dupes"]))
    a, b = sg._get_tf_idf_matrices()
    np.testing.assert_array_equal(a.toarray(), np.array(ka["tfidf_master_dupes"]["master_dense"]))
    np.testing.assert_array_equal(b.toarray(), np.array(ka["tfidf_master_dupes"]["dupes_dense"]))
    np.testing.assert_array_equal(sg._build_matches(a, b, None).toarray(), np.array(ka["build_matches_3x3"]["dense"]))
    df = api.StringGrouper(pd.Series(ka["get_matches_single"]["input"])).fit().get_matches()
    assert df.left_index.tolist() == ka["get_matches_single"]["left_index"]
    assert df.right_index.tolist() == ka["get_matches_single"]["right_index"]
    df = api.match_strings(pd.Series(ka["zero_min_similarity"]["master"], name="Customer Name"),
                           pd.Series(ka["zero_min_similarity"]["dupes"]), min_similarity=0)
    assert abs(df.similarity[0] - ka["zero_min_similarity"]["score_row1"]) < 5e-9 and df.left_index[0] == 1
    assert (df.similarity[1:] == 0).all() and len(df) == 6
    sims = api.compute_pairwise_similarities(cust, pd.Series(ka["centroid_groups"]["value"]))
    np.testing.assert_allclose(sims.to_numpy(), np.array(ka["pairwise"]["values"]), rtol=1e-12)
    np.testing.assert_array_equal(sims.to_numpy(), np.array([float.fromhex(x) for x in CASES["pairwise_customers"]["values_hex"]]))
    df = api.match_strings(acc, min_similarity=0.7)
    nonself = df[df.left_index != df.right_index]
    assert sorted(set(nonself.left_index)) == ka["accounts_ids_with_matches_at_07"]["rows"]
    assert api.group_similar_strings(cust, min_similarity=0.6, ignore_index=True).tolist() == ka["centroid_groups"]["value"]
    assert api.group_similar_strings(cust, min_similarity=0.6, ignore_index=True, group_rep="first").tolist() == \
        ka["first_groups"]["value"]
