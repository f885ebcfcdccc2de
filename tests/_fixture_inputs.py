"""Inputs of the reference-made fixtures at scale (tests/golden/make_golden_large.py makes the fixtures by
running the UNMODIFIED reference on exactly these inputs; the GPU tests rebuild the inputs from the seeds).
Deterministic: numpy Generator with fixed seeds + the SynthNames-v1 generator."""
from __future__ import annotations

import numpy as np

from string_grouper_amd.synth import synth_names

# rows the synthetic generator never produces: hubs of identical names (ties at the top-n cut), a chain of
# near-duplicates (connected component wider than any single match), empty / shorter-than-an-n-gram rows,
# rows that consist of deleted characters only, non-ASCII rows (lower + NFKD on the way to ASCII)
_SPECIAL = (["ACME HOLDINGS INC"] * 180 + ["GLOBAL VENTURE PARTNERS LLC"] * 37 +
            ["DELTA RIVER TRUST", "DELTA RIVER TRUST CO", "DELTA RIVER TRUST COMPANY", "THE DELTA RIVER TRUST COMPANY",
             "THE DELTA RIVER TRUST COMPANY LTD", "DELTA RIVERS TRUST COMPANY LTD"] +
            ["", "A", "AB", "  ", ".,-/", "A.B", "ABC", "abc", "Abc Inc", "abc inc."] +
            ["Ünïcödé Straße GmbH", "Unicode Strasse GmbH", "CAFÉ DU MONDE LLC", "CAFE DU MONDE LLC",
             "Crème Brûlée Holdings", "CREME BRULEE HOLDINGS", "ÀbracâDABRÀ", "Łódź Fabryka SA", "東京 Holdings KK"])


def fixture_names(n: int, seed: int) -> list:
    rng = np.random.default_rng(seed + 991)
    names = synth_names(n - len(_SPECIAL), seed) + list(_SPECIAL)
    order = rng.permutation(len(names))
    return [names[i] for i in order]


def fixture_master_and_duplicates(n_master: int, n_dup: int, seed: int):
    """Duplicates: half of them perturbed master entries (the synthetic generator's rule), half fresh names."""
    master = fixture_names(n_master, seed)
    dups = synth_names(n_dup - 12, seed + 17, perturb_of=master, perturb_frac=0.5) + \
        ["", "AB", "ACME HOLDINGS INC", "ACME HOLDINGS", "acme holdings inc.", "DELTA RIVER TRUST CO.", "nothing alike at all",
         "Ünïcödé Straße", "CAFÉ DU MONDE", "GLOBAL VENTURE PARTNERS", "GLOBAL VENTURE PARTNERS LLC", "ZZZZZZ"]
    rng = np.random.default_rng(seed + 5)
    order = rng.permutation(len(dups))
    return master, [dups[i] for i in order]


# the cases: name -> (kind, input spec, kwargs).  dtype given by name to keep this importable without numpy dtypes
CASES = {
    "selfjoin_30k_f32": ("match_strings", ("self", 30000, 20240901), dict(max_n_matches=10, min_similarity=0.8, tfidf_matrix_dtype="float32")),
    "selfjoin_30k_f64": ("match_strings", ("self", 30000, 20240901), dict(max_n_matches=10, min_similarity=0.8)),
    "selfjoin_20k_default": ("match_strings", ("self", 20000, 20240902), dict()),
    "selfjoin_20k_low_threshold_top3": ("match_strings", ("self", 20000, 20240902), dict(max_n_matches=3, min_similarity=0.6, tfidf_matrix_dtype="float32")),
    "master_dups_20k_x_8k_f32": ("match_strings", ("pair", 20000, 8000, 20240903), dict(max_n_matches=20, min_similarity=0.7, tfidf_matrix_dtype="float32")),
    "master_dups_20k_x_8k_f64": ("match_strings", ("pair", 20000, 8000, 20240903), dict(max_n_matches=20, min_similarity=0.7)),
    "groups_30k_centroid_f64": ("group_similar_strings", ("self", 30000, 20240901), dict(min_similarity=0.8)),
    "groups_30k_first_f32": ("group_similar_strings", ("self", 30000, 20240901), dict(min_similarity=0.8, group_rep="first", tfidf_matrix_dtype="float32")),
    "groups_20k_centroid_f32_low": ("group_similar_strings", ("self", 20000, 20240902), dict(min_similarity=0.7, tfidf_matrix_dtype="float32")),
    "most_similar_20k_x_8k_f64": ("match_most_similar", ("pair", 20000, 8000, 20240903), dict(min_similarity=0.7)),
    "most_similar_20k_x_8k_f32": ("match_most_similar", ("pair", 20000, 8000, 20240903), dict(min_similarity=0.6, tfidf_matrix_dtype="float32")),
}


def build_inputs(spec):
    if spec[0] == "self":
        return fixture_names(spec[1], spec[2]), None
    return fixture_master_and_duplicates(spec[1], spec[2], spec[3])


def resolve_kwargs(kw):
    kw = dict(kw)
    if "tfidf_matrix_dtype" in kw:
        kw["tfidf_matrix_dtype"] = getattr(np, kw["tfidf_matrix_dtype"])
    return kw
