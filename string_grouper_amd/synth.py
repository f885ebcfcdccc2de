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


def _row_key(name: str) -> str:
    """What decides a name's TF-IDF row under the reference's defaults (lower case, regex '[,-./]|\\s' deleted,
    string_grouper.py:190-197): two names of one key have identical rows."""
    return "".join(c for c in name.lower() if c not in ",-./" and not c.isspace())


def synth_names(n: int, seed: int = 1234, perturb_of: Optional[Sequence[str]] = None,
                perturb_frac: float = 0.30, dup_frac: Optional[float] = None) -> List[str]:
    """Generate ``n`` names.  ``perturb_of`` given: perturbations are drawn from that list
    (used for the 'duplicates' side of config 5 with perturb_frac=0.5) instead of from the
    names generated so far.

    ``dup_frac`` (round 6): SynthNames-v1 repeats itself far more than a real company list -- 16.5 % of its names have the TF-IDF
    row of an earlier name (exact copies, punctuation variants, one-word names drawn twice), where the reference's README
    counts 1 747 names in groups of identical names among 663 000 (0.26 %).  With ``dup_frac`` given only the first
    ``dup_frac * n`` such repeats stay; every later one is replaced by a FRESH name (words drawn from the same lexicon by a
    stream of its own, ``seed + 7919``) whose row no earlier name has.  ``None``: the survey's generator as it is."""
    if dup_frac is not None:
        base = synth_names(n, seed, perturb_of, perturb_frac)
        return _thin_repeats(base, seed, float(dup_frac))
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


def _thin_repeats(names: List[str], seed: int, dup_frac: float) -> List[str]:
    rng = np.random.default_rng(seed + 7919)
    lex = _lexicon(np.random.default_rng(seed))           # (the list's own lexicon: the first thing its stream draws)
    ranks = np.arange(_LEXICON_SIZE, dtype=np.float64)
    pw = 1.0 / (ranks + 5.0) ** 0.95
    pw /= pw.sum()
    suf_names = [s for s, _ in _SUFFIXES]
    sw = np.array([w for _, w in _SUFFIXES], dtype=np.float64)
    sw /= sw.sum()
    keep = int(dup_frac * len(names))
    seen, out, kept = set(), list(names), 0
    redo = []
    for i, name in enumerate(names):
        key = _row_key(name)
        if key in seen:
            if kept < keep:
                kept += 1
            else:
                redo.append(i)
        else:
            seen.add(key)
    # (drawn in bulk: a draw from the 20 000-word law per name is a second per thousand names)
    m = int(len(redo) * 1.3) + 64
    ks = rng.choice(np.array([2, 3, 4]), size=m, p=[0.5, 0.35, 0.15])             # (one-word names are what collides)
    words = rng.choice(_LEXICON_SIZE, size=int(ks.sum()), p=pw)
    has_suffix = rng.random(m) < 0.8
    suffix_ids = rng.choice(len(suf_names), size=m, p=sw)
    starts = np.concatenate([[0], np.cumsum(ks)])
    at = 0
    for i in redo:
        while True:
            if at >= m:
                raise RuntimeError("synth_names(dup_frac=...): ran out of fresh names")
            parts = [lex[w] for w in words[starts[at]:starts[at + 1]]]
            if has_suffix[at]:
                parts.append(suf_names[int(suffix_ids[at])])
            at += 1
            cand = " ".join(parts)
            key = _row_key(cand)
            if key not in seen:
                seen.add(key)
                out[i] = cand
                break
    return out
