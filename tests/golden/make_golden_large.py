"""Generates tests/golden/reference_large.npz by running the UNMODIFIED reference package
(/root/reference/string_grouper, v0.7.1) on the inputs of tests/_fixture_inputs.py: 20-30 k names with hubs of
identical names, chains of near-duplicates, empty / short / non-ASCII rows; self-joins and master x duplicates;
float32 and float64; match_strings (string_grouper.py:130-153 -> fit :380-431 -> get_matches :442-500),
group_similar_strings (:70-92 -> _deduplicate :851-904, both group_rep values) and match_most_similar
(:95-127 -> _get_nearest_matches :783-849).

The reference needs ``sparse_dot_topn`` (third party, absent here, source not in the reference tree); it is
substituted by tests/ref_shims/sparse_dot_topn with the C restatement oracle/sdtn_port.c (SG_SHIM_BACKEND=port),
i.e. everything except that one function is the reference's own code: the n-gram analyzer, sklearn's
TfidfVectorizer, the block split, vstack, the lil-matrix symmetrisation, pandas merges, connected components.
Where several candidates tie at the top-n cut the fixture carries the restatement's rule (score descending,
column ascending) -- the one point the reference's own tests do not pin.

What is stored (compressed npz, a few MB): per case the result reduced to integer / float64 arrays --
match_strings: left_index, right_index, similarity (float64 bits); groups / most similar: the POSITION in the
input of the string each row was mapped to (the returned strings are looked up by position, ties between equal
strings resolved by the reference's own index columns).

Run:  python tests/golden/make_golden_large.py      (needs /root/reference; ~2 min; not run on the GPU box)
"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
os.environ["SG_SHIM_BACKEND"] = "port"
sys.path.insert(0, ROOT)                                     # for ``oracle`` (used by the shims) and tests/_fixture_inputs
sys.path.insert(0, os.path.join(ROOT, "tests", "ref_shims"))
sys.path.insert(0, "/root/reference")                        # must win over the repo's own drop-in alias

import numpy as np  # noqa: E402
import pandas as pd  # noqa: E402
import string_grouper as ref  # noqa: E402

assert "/root/reference" in ref.__file__, "must import the reference package"
from tests import _fixture_inputs as F  # noqa: E402

out = {}
for name, (kind, spec, kw) in F.CASES.items():
    t0 = time.time()
    master, dups = F.build_inputs(spec)
    kwargs = F.resolve_kwargs(kw)
    m = pd.Series(master, name="name")
    d = None if dups is None else pd.Series(dups, name="dup")
    if kind == "match_strings":
        df = ref.match_strings(m, d, **kwargs)
        out[name + "/left_index"] = df["left_index"].to_numpy(dtype=np.int32)
        out[name + "/right_index"] = df["right_index"].to_numpy(dtype=np.int32)
        out[name + "/similarity"] = df["similarity"].to_numpy(dtype=np.float64)
        n_out = len(df)
    elif kind == "group_similar_strings":
        g = ref.group_similar_strings(m, **kwargs)          # frame: group_rep_index, group_rep_name
        out[name + "/group_rep_index"] = g["group_rep_index"].to_numpy(dtype=np.int32)
        assert (g["group_rep_name"].to_numpy() == m.to_numpy()[g["group_rep_index"].to_numpy()]).all()
        n_out = int(g["group_rep_index"].nunique())
    else:
        r = ref.match_most_similar(m, d, **kwargs)          # frame: most_similar_index, most_similar_name
        idx = r["most_similar_index"].to_numpy()
        strs = r["most_similar_name"].to_numpy()
        # unmatched duplicates keep their own string and (ignore_index=False, replace_na=False) a NaN index
        matched = ~pd.isna(idx)
        pos = np.full(len(idx), -1, np.int32)
        pos[matched] = idx[matched].astype(np.int32)
        assert (strs[matched] == m.to_numpy()[pos[matched]]).all() and (strs[~matched] == d.to_numpy()[~matched]).all()
        out[name + "/most_similar_index"] = pos
        n_out = int(matched.sum())
    print(f"{name:40s} {kind:22s} -> {n_out:7d}   {time.time() - t0:6.1f} s", flush=True)

path = os.path.join(HERE, "reference_large.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path) // 1024, "KiB")
