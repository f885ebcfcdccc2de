"""Generates tests/golden/reference_golden.json by running the UNMODIFIED reference package
(/root/reference/string_grouper, v0.7.1) in this container.

The reference needs two packages that are absent here: ``sparse_dot_topn`` (third party, source not
in the reference tree) and ``loguru``.  They are substituted by tests/ref_shims/ (oracle-backed
sp_matmul_topn / zip_sp_matmul_topn; a logging stand-in).  Consequently:
  * "tfidf" entries are produced by the reference's own code + sklearn (fully reference-real);
  * "matches"/"groups"/"most_similar" entries pass through the oracle for the sparse top-n multiply
    and are only as pinned as the oracle is (see oracle/oracle.py header) -- on these tiny inputs
    there are no ties at the cut and no score at the threshold, so they are unambiguous.
Known answers hard-coded in the reference's tests are transcribed under "known_answers" with
their file:line.

Run:  python tests/golden/make_golden.py      (needs /root/reference; not run on the GPU box)
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)                                     # for ``oracle`` (used by the shims)
sys.path.insert(0, os.path.join(ROOT, "tests", "ref_shims"))
sys.path.insert(0, "/root/reference")                        # must win over the repo's own drop-in alias

import numpy as np  # noqa: E402
import pandas as pd  # noqa: E402
from string_grouper import StringGrouper, match_strings, group_similar_strings, match_most_similar, \
    compute_pairwise_similarities  # noqa: E402

assert "/root/reference" in sys.modules["string_grouper"].__file__, "must import the reference package"

ACCOUNTS = [l.split(",", 1) for l in open("/root/reference/tutorials/accounts.csv").read().strip().split("\n")[1:]]
ACCOUNT_IDS = [a for a, _ in ACCOUNTS]
ACCOUNT_NAMES = [b for _, b in ACCOUNTS]
CUSTOMERS = ['Mega Enterprises Corporation', 'Hyper Startup Incorporated', 'Hyper Startup Inc.',
             'Hyper-Startup Inc.', 'Hyper Hyper Inc.', 'Mega Enterprises Corp.']            # test:21-26
CUSTOMERS2 = CUSTOMERS[:4] + ['HyperStartup Inc.'] + CUSTOMERS[4:]                          # test:32-38


def frame(df):
    df = df.reset_index(drop=True)
    return {"columns": list(map(str, df.columns)), "rows": [[(x.item() if hasattr(x, "item") else x) for x in row]
                                                           for row in df.itertuples(index=False, name=None)]}


def csr(m):
    m = m.tocsr()
    m.sort_indices()
    return {"shape": list(m.shape), "indptr": m.indptr.tolist(), "indices": m.indices.tolist(),
            "data_hex": [float(x).hex() for x in m.data.astype(np.float64)], "dtype": str(m.dtype)}


out = {"generator": "tests/golden/make_golden.py", "reference": "Bergvca/string_grouper v0.7.1 (/root/reference)",
       "inputs": {"accounts_names": ACCOUNT_NAMES, "accounts_ids": ACCOUNT_IDS, "customers": CUSTOMERS,
                  "customers2": CUSTOMERS2}, "cases": {}}
cases = out["cases"]

# ---- TF-IDF matrices straight from the reference + sklearn
for name, series, kw in [("accounts_f64", ACCOUNT_NAMES, {}), ("accounts_f32", ACCOUNT_NAMES, {"tfidf_matrix_dtype": np.float32}),
                         ("customers_f64", CUSTOMERS, {}), ("customers_case", CUSTOMERS, {"ignore_case": False}),
                         ("customers_ngram2", CUSTOMERS, {"ngram_size": 2})]:
    sg = StringGrouper(pd.Series(series), **kw)
    m, _ = sg._get_tf_idf_matrices()
    vocab = sg._vectorizer.vocabulary_
    cases["tfidf_" + name] = {"kwargs": {k: (v.__name__ if isinstance(v, type) else v) for k, v in kw.items()},
                              "input": "accounts_names" if series is ACCOUNT_NAMES else "customers",
                              "matrix": csr(m), "vocabulary": sorted(vocab, key=vocab.get)}
sg = StringGrouper(pd.Series(CUSTOMERS), pd.Series(CUSTOMERS2))
a, b = sg._get_tf_idf_matrices()
cases["tfidf_customers_vs_customers2"] = {"master": csr(a), "duplicates": csr(b)}

# ---- match lists / frames (oracle-backed multiply)
for name, kw in [("accounts_default", {}), ("accounts_min07", {"min_similarity": 0.7}),
                 ("accounts_top2", {"max_n_matches": 2, "min_similarity": 0.5}),
                 ("accounts_f32", {"tfidf_matrix_dtype": np.float32}),
                 ("accounts_blocks_2_3", {"n_blocks": (2, 3), "min_similarity": 0.5})]:
    df = match_strings(pd.Series(ACCOUNT_NAMES, name="name"), **kw)
    cases["match_strings_" + name] = {"kwargs": {k: (v.__name__ if isinstance(v, type) else v) for k, v in kw.items()},
                                      "frame": frame(df)}
df = match_strings(pd.Series(CUSTOMERS, name="Customer Name"), pd.Series(CUSTOMERS2, name="Customer Name"),
                   min_similarity=0.1)
cases["match_strings_customers_vs_customers2_min01"] = {"frame": frame(df)}
df = match_strings(pd.Series(ACCOUNT_NAMES, name="name"), master_id=pd.Series(ACCOUNT_IDS, name="id"), min_similarity=0.7)
cases["match_strings_accounts_with_ids_min07"] = {"frame": frame(df)}

g = group_similar_strings(pd.Series(ACCOUNT_NAMES, name="name"), min_similarity=0.7)
cases["group_accounts_min07"] = {"frame": frame(g)}
g = group_similar_strings(pd.Series(ACCOUNT_NAMES, name="name"), min_similarity=0.7, group_rep="first", ignore_index=True)
cases["group_accounts_min07_first"] = {"values": g.tolist(), "name": g.name}
m = match_most_similar(pd.Series(CUSTOMERS, name="Customer Name"), pd.Series(CUSTOMERS2 + ["nothing alike"], name="dup"),
                       min_similarity=0.6)
cases["most_similar_customers"] = {"frame": frame(m)}
p = compute_pairwise_similarities(pd.Series(CUSTOMERS), pd.Series(['Mega Enterprises Corporation', 'Hyper Startup Inc.',
                                  'Hyper Startup Inc.', 'Hyper Startup Inc.', 'Hyper Hyper Inc.', 'Mega Enterprises Corporation']))
cases["pairwise_customers"] = {"values_hex": [float(x).hex() for x in p.values]}

# ---- known answers transcribed from the reference's own tests / docs
out["known_answers"] = {
    "ngrams_McDonalds_case": {"src": "string_grouper/test/test_string_grouper.py:495-501",
                              "value": ['McD', 'cDo', 'Don', 'ona', 'nal', 'ald', 'lds']},
    "ngrams_McDonalds_lower": {"src": "string_grouper/test/test_string_grouper.py:503-517",
                               "value": ['mcd', 'cdo', 'don', 'ona', 'nal', 'ald', 'lds']},
    "ngrams_unicode": {"src": "docs/references/sg_class.md:54-57", "input": "ÀbracâDABRÀ",
                       "value": ['abr', 'bra', 'rac', 'aca', 'cad', 'ada', 'dab', 'abr', 'bra']},
    "tfidf_foo_bar_baz": {"src": "string_grouper/test/test_string_grouper.py:519-528",
                          "input": ['foo', 'bar', 'baz'], "dense": [[0., 0., 1.], [1., 0., 0.], [0., 1., 0.]]},
    "tfidf_master_dupes": {"src": "string_grouper/test/test_string_grouper.py:530-544",
                           "master": ['foo', 'bar', 'baz'], "dupes": ['foo', 'bar', 'bop'],
                           "master_dense": [[0., 0., 0., 1.], [1., 0., 0., 0.], [0., 1., 0., 0.]],
                           "dupes_dense": [[0., 0., 0., 1.], [1., 0., 0., 0.], [0., 0., 1., 0.]]},
    "build_matches_3x3": {"src": "string_grouper/test/test_string_grouper.py:546-556",
                          "dense": [[1., 0., 0.], [0., 1., 0.], [0., 0., 0.]]},
    "get_matches_single": {"src": "string_grouper/test/test_string_grouper.py:599-612", "input": ['foo', 'bar', 'baz', 'foo'],
                           "left_index": [0, 0, 1, 2, 3, 3], "right_index": [0, 3, 1, 2, 0, 3]},
    "zero_min_similarity": {"src": "string_grouper/test/test_string_grouper.py:46-56,478-485",
                            "master": CUSTOMERS, "dupes": ['whatever'], "score_row1": 0.08170638},
    "pairwise": {"src": "string_grouper/test/test_string_grouper.py:364-382",
                 "values": [1.0, 0.6336195351561589, 1.0000000000000004, 1.0000000000000004, 1.0, 0.826462625999832]},
    "accounts_ids_with_matches_at_07": {"src": "tutorials/tutorial_1.md:420-432",
                                        "rows": [0, 1, 3, 4, 5, 7, 8, 10, 11, 12]},
    "centroid_groups": {"src": "string_grouper/test/test_string_grouper.py:57-67,684-696", "min_similarity": 0.6,
                        "value": ['Mega Enterprises Corporation', 'Hyper Startup Inc.', 'Hyper Startup Inc.',
                                  'Hyper Startup Inc.', 'Hyper Hyper Inc.', 'Mega Enterprises Corporation']},
    "first_groups": {"src": "string_grouper/test/test_string_grouper.py:79-89,767-780", "min_similarity": 0.6,
                     "value": ['Mega Enterprises Corporation', 'Hyper Startup Incorporated', 'Hyper Startup Incorporated',
                               'Hyper Startup Incorporated', 'Hyper Hyper Inc.', 'Mega Enterprises Corporation']},
}
with open(os.path.join(HERE, "reference_golden.json"), "w") as f:
    json.dump(out, f, indent=1, ensure_ascii=False)
print("wrote", os.path.join(HERE, "reference_golden.json"), len(cases), "cases")
