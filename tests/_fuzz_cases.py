"""Seeded random jobs for the public API (tests/test_host_api.py: against the unmodified reference on CPU;
tests/test_parity_gpu.py: the HIP engine against the oracle engine).  TEST INFRASTRUCTURE ONLY."""
import numpy as np, pandas as pd
from string_grouper_amd.synth import synth_names


def cases(n_cases=150, seed=20260926):
    rng = np.random.default_rng(seed)
    pool = synth_names(400, 3) + ["", "A", "AB", "ÉCOLE NORMALE", "Straße 5 GmbH", "ΣΊΣΥΦΟΣ ΑΕ", "x" * 70, "O'NEIL & SONS, LTD."]
    for c in range(n_cases):
        n = int(rng.integers(3, 60))
        master = list(rng.choice(pool, n))
        kw = {}
        if rng.random() < 0.5:
            kw["min_similarity"] = float(rng.choice([0.3, 0.5, 0.7, 0.8, 0.95]))
        if rng.random() < 0.5:
            kw["max_n_matches"] = int(rng.choice([1, 2, 5, 20, n]))
        if rng.random() < 0.4:
            kw["ngram_size"] = int(rng.choice([2, 3, 4, 5]))
        if rng.random() < 0.3:
            kw["ignore_case"] = bool(rng.random() < 0.5)
        if rng.random() < 0.3:
            kw["normalize_to_ascii"] = bool(rng.random() < 0.5)
        if rng.random() < 0.3:
            kw["regex"] = str(rng.choice([r"[,-./]|\s", r"[^A-Za-z0-9 ]", r"(INC|LLC)", r"\d+"]))
        if rng.random() < 0.5:
            kw["tfidf_matrix_dtype"] = np.float32 if rng.random() < 0.5 else np.float64
        if rng.random() < 0.3:
            kw["n_blocks"] = (int(rng.integers(1, 4)), int(rng.integers(1, 4)))
        if rng.random() < 0.3:
            kw["ignore_index"] = bool(rng.random() < 0.5)
        kind = str(rng.choice(["match", "match_dup", "groups", "nearest", "pairwise"]))
        ids = rng.random() < 0.4
        m = pd.Series(master, name=None if rng.random() < 0.5 else "nm")
        mid = pd.Series(np.arange(n) * 3 + 7, name="uid") if ids else None
        d = did = None
        if kind in ("match_dup", "nearest"):
            k = int(rng.integers(2, 30))
            d = pd.Series(list(rng.choice(pool, k)), name=None if rng.random() < 0.5 else "dp")
            did = pd.Series([f"d{i}" for i in range(k)]) if ids else None
        if kind == "groups":
            if rng.random() < 0.5:
                kw["group_rep"] = str(rng.choice(["centroid", "first"]))
            if rng.random() < 0.3:
                kw["replace_na"] = False
        if kind == "nearest" and rng.random() < 0.3 and ids:
            kw["replace_na"] = bool(rng.random() < 0.5)
        if kind == "pairwise":
            d = pd.Series(list(rng.choice(pool, n)))
            kw = {k_: v for k_, v in kw.items() if k_ in ("ngram_size", "ignore_case", "normalize_to_ascii", "regex", "tfidf_matrix_dtype")}
        yield c, kind, m, d, mid, did, kw


def run(api, kind, m, d, mid, did, kw):
    try:
        if kind in ("match", "match_dup"):
            return api.match_strings(m, d, mid, did, **kw)
        if kind == "groups":
            return api.group_similar_strings(m, mid, **kw)
        if kind == "nearest":
            return api.match_most_similar(m, d, mid, did, **kw)
        return api.compute_pairwise_similarities(m, d, **kw)
    except Exception as e:                     # the same inputs must fail the same way
        return ("raised", type(e).__name__, str(e)[:80])
