"""SynthNames-v1: deterministic synthetic company-name generator.

Stands in for the sec__edgar company list (not distributable, not in the reference tree;
linked only from /root/reference/docs/performance.md:67).  Upper-case ASCII names built from a
Zipf-distributed pseudo-word lexicon plus legal suffixes, 30 % of them near-duplicates of an
earlier name, so that the TF-IDF matrix has the skew that matters for the sparse top-n multiply
(a few n-grams such as 'inc'/'llc' present in >10 % of the rows).
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np

_CONS = "BCDFGHJKLMNPRSTVWZ"
_VOWS = "AEIOU"
_SUFFIXES = [("INC", 18), ("LLC", 16), ("CORP", 10), ("LTD", 7), ("LP", 6), ("CO", 5), ("TRUST", 5),
             ("FUND", 4), ("HOLDINGS", 4), ("GROUP", 4), ("PARTNERS", 3), ("CAPITAL", 3),
             ("INCORPORATED", 2), ("CORPORATION", 2), ("LIMITED", 2), ("COMPANY", 2), ("/DE/", 1),
             ("/BD", 1), ("& CO", 2), ("INTERNATIONAL", 3)]
_LEXICON_SIZE = 20000


def _lexicon(rng: np.random.Generator) -> List[str]:
    words, seen = [], set()
    while len(words) < _LEXICON_SIZE:
        nsyl = int(rng.integers(2, 5))
        parts = []
        for _ in range(nsyl):
            parts.append(_CONS[int(rng.integers(len(_CONS)))])
            parts.append(_VOWS[int(rng.integers(len(_VOWS)))])
            if rng.random() < 0.4:
                parts.append(_CONS[int(rng.integers(len(_CONS)))])
        w = "".join(parts)
        if w not in seen:
            seen.add(w)
            words.append(w)
    order = rng.permutation(len(words))
    return [words[i] for i in order]


def _perturb(name: str, kind: int, u1: float, u2: float) -> str:
    if kind == 0 and len(name) > 1:                       # delete one character
        p = int(u1 * len(name))
        return name[:p] + name[p + 1:]
    if kind == 1 and len(name) > 0:                       # substitute one character
        p = int(u1 * len(name))
        return name[:p] + _CONS[int(u2 * len(_CONS))] + name[p + 1:]
    if kind == 2:                                         # punctuation variant
        sp_ = name.find(" ")
        if sp_ > 0:
            return name[:sp_] + ", " + name[sp_ + 1:] + "."
        return name + "."
    if kind == 3:                                         # suffix long form
        if name.endswith(" INC"):
            return name[:-3] + "INCORPORATED"
        if name.endswith(" CORP"):
            return name[:-4] + "CORPORATION"
        return name + " INC"
    if kind == 4:                                         # exact duplicate
        return name
    return name + " " + str(1 + int(u1 * 29))            # numbered series


def synth_names(n: int, seed: int = 1234, perturb_of: Optional[Sequence[str]] = None,
                perturb_frac: float = 0.30) -> List[str]:
    """Generate ``n`` names.  ``perturb_of`` given: perturbations are drawn from that list
    (used for the 'duplicates' side of config 5 with perturb_frac=0.5) instead of from the
    names generated so far."""
    rng = np.random.default_rng(seed)
    lex = _lexicon(rng)
    ranks = np.arange(_LEXICON_SIZE, dtype=np.float64)
    pw = 1.0 / (ranks + 5.0) ** 0.95
    pw /= pw.sum()
    suf_names = [s for s, _ in _SUFFIXES]
    sw = np.array([w for _, w in _SUFFIXES], dtype=np.float64)
    sw /= sw.sum()

    nwords = rng.choice(np.array([1, 2, 3, 4]), size=n, p=[0.15, 0.45, 0.30, 0.10])
    word_ids = rng.choice(_LEXICON_SIZE, size=int(nwords.sum()), p=pw)
    has_suffix = rng.random(n) < 0.8
    suffix_ids = rng.choice(len(suf_names), size=n, p=sw)
    is_pert = rng.random(n) < perturb_frac
    pert_kind = rng.choice(6, size=n, p=[0.25, 0.20, 0.20, 0.15, 0.10, 0.10])
    pert_src = rng.random(n)
    u1 = rng.random(n)
    u2 = rng.random(n)

    out: List[str] = []
    wpos = 0
    for i in range(n):
        k = int(nwords[i])
        if is_pert[i] and (perturb_of is not None or i >= 100):
            pool = perturb_of if perturb_of is not None else out
            src = pool[int(pert_src[i] * len(pool))]
            out.append(_perturb(src, int(pert_kind[i]), float(u1[i]), float(u2[i])))
        else:
            parts = [lex[w] for w in word_ids[wpos:wpos + k]]
            if has_suffix[i]:
                parts.append(suf_names[int(suffix_ids[i])])
            out.append(" ".join(parts))
        wpos += k
    return out
