"""The host string preparation (string_grouper_amd/strprep.py) against the reference's analyzer restated in
oracle.ngrams (string_grouper.py:365-378): for every option combination the n-grams the DEVICE tokeniser would
produce from the prepared column (its byte-level rules emulated here in numpy) must equal the reference's, string
by string -- incl. '™' (NFKD gives upper-case ASCII after lower()), 'İ', 'ß', final sigma, ligatures, full-width
forms, combining marks, CJK, astral code points, empty and all-deleted rows."""
import re

import numpy as np
import pytest

from oracle import oracle as O
from string_grouper_amd import strprep as SP

CORPUS = ["", "a", "ab", "abc", "ABC Inc.", "Ünïcödé Straße GmbH", "™ TRADEMARK Co", "№ 5 ℡ 12 ㎆", "İstanbul ISTANBUL ıi",
          "ΟΔΥΣΣΕΥΣ ΣΟΦΟΣ Σ", "ὈΔΥΣΣΕΎΣ", "ﬁne ﬂour ﬃ Ⅻ ½", "ＦＵＬＬ　ｗｉｄｔｈ", "é å ȫ", "東京 Holdings 株式会社",
          "😀 emoji 𝔘𝔫𝔦 𝒜", "  ..,,--//  ", "tab\there\nnew\x1cfs", "ÀbracâDABRÀ", "Crème Brûlée", "Łódź", "ǅ ǆ Ǆ", "ẞ ß SS",
          "a.b,c-d/e f", "ẛ̣", "Å Å Å", "Ω Ω", "x" * 70 + "é" + "y" * 70, "mixed ASCII and ünï", "ALL ASCII HERE 123"]


def fuzz_corpus(n, seed):
    rng = np.random.default_rng(seed)
    pools = [list(range(0x20, 0x7F)), list(range(0xA0, 0x250)), list(range(0x370, 0x400)), list(range(0x400, 0x460)),
             list(range(0x1E00, 0x1F00)), list(range(0x2100, 0x2190)), list(range(0xFB00, 0xFB07)), list(range(0xFF01, 0xFF5F)),
             list(range(0x300, 0x340)), list(range(0x4E00, 0x4E40)), list(range(0x1D400, 0x1D440)), [0x3A3, 0x130, 0x131, 0x2122]]
    out = []
    for _ in range(n):
        k = int(rng.integers(0, 24))
        chars = []
        for _ in range(k):
            pool = pools[0] if rng.random() < 0.6 else pools[int(rng.integers(len(pools)))]
            chars.append(chr(pool[int(rng.integers(len(pool)))]))
        out.append("".join(chars))
    return out


def device_ngrams(col, ngram_size, ignore_case, regex):
    """What K1 makes of a prepared column (sg_vectorize.hip): bytes -> drop >= 0x80, lower A-Z unless prelowered, drop the
    delete table; symbols -> as they are; then all n-grams."""
    table = SP.delete_table_for(regex) if SP.regex_is_char_class(regex) else np.zeros(128, np.uint8)
    if col.kind == "bytes" and not (SP.regex_is_char_class(regex)):
        table = np.zeros(128, np.uint8)
    out = []
    for i in range(col.n):
        seg = col.data[col.offsets[i]:col.offsets[i + 1]]
        if col.kind == "bytes":
            seg = seg[seg < 0x80]
            if ignore_case and not col.prelowered:
                seg = np.where((seg >= 65) & (seg <= 90), seg + 32, seg)
            seg = seg[table[seg] == 0]
        s = "".join(chr(int(c)) for c in seg)
        out.append([s[j:j + ngram_size] for j in range(len(s) - ngram_size + 1)])
    return out


@pytest.mark.parametrize("ignore_case", [True, False])
@pytest.mark.parametrize("normalize_to_ascii", [True, False])
@pytest.mark.parametrize("regex", [O.DEFAULT_REGEX, r"[aeiouéß]", r"inc\.?|\s", r"\W", r"[^a-z]", r"(?i)co"])
def test_prepared_column_gives_the_reference_ngrams(ignore_case, normalize_to_ascii, regex):
    strings = CORPUS + fuzz_corpus(400, 7)
    col = SP.prepare_column(np.asarray(strings, dtype=object), ignore_case, normalize_to_ascii, regex)
    for n in (2, 3):
        got = device_ngrams(col, n, ignore_case, regex)
        for s, g in zip(strings, got):
            want = O.ngrams(s, n, regex, ignore_case, normalize_to_ascii)
            assert g == want, (s, ignore_case, normalize_to_ascii, regex, n)


def test_ascii_columns_are_left_to_the_device():
    col = SP.prepare_column(np.asarray(["ACME Inc.", "foo-bar"], dtype=object), True, True, O.DEFAULT_REGEX)
    assert col.kind == "bytes" and not col.prelowered and bytes(col.data) == b"ACME Inc.foo-bar"


def test_many_non_ascii_rows_take_the_gather_path():
    strings = [("é" if i % 2 else "") + f"name {i} Ü" for i in range(9000)]       # > 4096 touched rows
    col = SP.prepare_column(np.asarray(strings, dtype=object), True, True, O.DEFAULT_REGEX)
    got = device_ngrams(col, 3, True, O.DEFAULT_REGEX)
    assert all(g == O.ngrams(s) for s, g in zip(strings, got))


def test_bytes_column_as_symbols():
    col = SP.prepare_column(np.asarray(["ACME, Inc.", "a-b c"], dtype=object), True, False, O.DEFAULT_REGEX)
    sym = SP.bytes_column_to_symbols(col, True, SP.delete_table_for(O.DEFAULT_REGEX))
    assert "".join(map(chr, sym.data)) == "acmeincabc" and sym.offsets.tolist() == [0, 7, 10]
