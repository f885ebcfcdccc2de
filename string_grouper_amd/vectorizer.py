"""Seam b1: a TfidfVectorizer-shaped front end of the device vectoriser (K1 + K2).

Mirrors how the reference builds and drives its vectoriser
(string_grouper/string_grouper.py:306 ``TfidfVectorizer(min_df=1, analyzer=self.n_grams, dtype=...)``,
``.fit`` at :706, ``.transform`` at :689/:692) and the analyzer ``StringGrouper.n_grams`` (:365-378).

Host work is limited to what the reference also does on the host per string and what has no
byte-level definition: Unicode ``str.lower()`` / NFKD for the (few) strings that contain non-ASCII
characters, and ``re.sub`` when the regex is not a plain character class.  Everything else --
ASCII lower-casing, character deletion, n-gramming, vocabulary, counts, idf weighting,
L2 normalisation -- runs on the GPU.  There is no CPU tokeniser fallback.
"""
from __future__ import annotations

import re
from typing import Dict, List, Optional, Sequence
from unicodedata import normalize as _ucd_normalize

import numpy as np

from . import _native as N

try:                                   # Python >= 3.11
    import re._parser as _sre_parse    # type: ignore
    import re._constants as _sre_c     # type: ignore
except ImportError:                    # Python 3.10
    import sre_parse as _sre_parse     # type: ignore
    import sre_constants as _sre_c     # type: ignore

DEFAULT_REGEX = r'[,-./]|\s'


def regex_is_char_class(pattern: str) -> bool:
    """True when every match of ``pattern`` is exactly one character chosen independently of
    context (a literal, a set, a category such as \\s, or an alternation of those), so that
    ``re.sub(pattern, '', s)`` deletes exactly the characters c with ``re.fullmatch(pattern, c)``."""
    try:
        parsed = _sre_parse.parse(pattern)
    except Exception:
        return False
    single = (_sre_c.LITERAL, _sre_c.NOT_LITERAL, _sre_c.IN, _sre_c.CATEGORY, _sre_c.ANY)

    def one_char(seq) -> bool:
        items = list(seq)
        if len(items) != 1:
            return False
        op, arg = items[0]
        if op in single:
            return True
        if op is _sre_c.BRANCH:
            return all(one_char(alt) for alt in arg[1])
        if op is _sre_c.SUBPATTERN:
            return one_char(arg[-1])
        return False

    if parsed.state.flags & (re.IGNORECASE | re.LOCALE):
        return False
    return one_char(parsed)


def delete_table_for(pattern: str) -> np.ndarray:
    rx = re.compile(pattern)
    return np.array([1 if rx.fullmatch(chr(c)) else 0 for c in range(128)], dtype=np.uint8)


class PreparedStrings:
    """A string column as Arrow large_string buffers (contiguous UTF-8 bytes + int64 offsets),
    after the host-side Unicode step; ``dev`` is its device-resident copy."""

    def __init__(self, data: np.ndarray, offsets: np.ndarray):
        self.data = data
        self.offsets = offsets
        self.n = len(offsets) - 1
        self.dev: Optional[N.Strings] = None


def _to_arrow_buffers(strings) -> (np.ndarray, np.ndarray):
    import pyarrow as pa
    if hasattr(strings, "array") and hasattr(strings.array, "_pa_array"):       # pandas ArrowExtensionArray
        arr = strings.array._pa_array.combine_chunks()
        arr = arr.cast(pa.large_string())
    else:
        values = strings.to_numpy() if hasattr(strings, "to_numpy") else np.asarray(strings, dtype=object)
        arr = pa.array(values, type=pa.large_string())
    if arr.null_count:
        raise TypeError("input contains null values; only strings are accepted")
    bufs = arr.buffers()
    offsets = np.frombuffer(bufs[1], dtype=np.int64, count=len(arr) + 1 + arr.offset)[arr.offset:]
    data = np.frombuffer(bufs[2], dtype=np.uint8) if bufs[2] is not None else np.zeros(0, np.uint8)
    if offsets[0] != 0:
        data = data[offsets[0]:offsets[-1]]
        offsets = offsets - offsets[0]
    else:
        data = data[:offsets[-1]]
    return data, np.ascontiguousarray(offsets)


def prepare_strings(strings, ignore_case: bool, normalize_to_ascii: bool, regex: Optional[str]) -> PreparedStrings:
    """Host step of n_grams (string_grouper.py:372-376) for what cannot be done bytewise.

    ``regex`` not None: apply ``re.sub(regex, '', s)`` on the host (pattern is not a character class)."""
    data, offsets = _to_arrow_buffers(strings)
    needs_unicode = data.size > 0 and int(data.max()) >= 0x80
    if not needs_unicode and regex is None:
        return PreparedStrings(data, offsets)
    # rows that need the Python string semantics
    if needs_unicode:
        hi = np.flatnonzero(data >= 0x80)
        rows = np.unique(np.searchsorted(offsets, hi, side="right") - 1)
        if not normalize_to_ascii:
            raise NotImplementedError(
                "normalize_to_ascii=False with non-ASCII input is not supported by the device tokeniser")
    else:
        rows = np.zeros(0, np.int64)
    values = list(strings)
    if regex is not None:
        rows = np.arange(len(values))
        rx = re.compile(regex)
    fixed = {}
    for r in rows:
        s = values[r]
        if ignore_case:
            s = s.lower()
        if normalize_to_ascii:
            s = _ucd_normalize('NFKD', s).encode('ASCII', 'ignore').decode()
        if regex is not None:
            s = rx.sub('', s)
        fixed[int(r)] = s
    for r, s in fixed.items():
        values[r] = s
    data, offsets = _to_arrow_buffers(np.asarray(values, dtype=object))
    if data.size and int(data.max()) >= 0x80:
        raise NotImplementedError("non-ASCII characters survive preprocessing; not supported on the device")
    return PreparedStrings(data, offsets)


def keys_to_terms(keys: np.ndarray, ngram_size: int) -> List[str]:
    out = []
    for k in keys.tolist():
        chars = [(k >> (7 * (ngram_size - 1 - q))) & 0x7F for q in range(ngram_size)]
        out.append(bytes(chars).decode('ascii'))
    return out


def idf_from_df(df: np.ndarray, n_docs: int, dtype) -> np.ndarray:
    """sklearn TfidfTransformer.fit op sequence (text.py:1664-1679) so that log() is bit-identical."""
    d = df.astype(dtype, copy=True)
    d += float(True)                   # smooth_idf
    idf = np.full_like(d, fill_value=n_docs + 1, dtype=dtype)
    idf /= d
    np.log(idf, out=idf)
    idf += 1.0
    return idf


class HipTfidfVectorizer:
    """fit / transform with the semantics of the reference's TfidfVectorizer instance.

    ``transform`` returns a scipy CSR (host) for the drop-in seam; ``transform_device`` keeps the
    matrix in HBM for the fused path."""

    def __init__(self, ngram_size: int = 3, regex: str = DEFAULT_REGEX, ignore_case: bool = True,
                 normalize_to_ascii: bool = True, dtype=np.float64, ctx: Optional[N.Context] = None):
        self.ngram_size = int(ngram_size)
        self.regex = regex
        self.ignore_case = bool(ignore_case)
        self.normalize_to_ascii = bool(normalize_to_ascii)
        self.dtype = np.dtype(dtype).type
        N.np_dtype_code(self.dtype)
        self._ctx = ctx
        self._vocab: Optional[N.Vocab] = None
        self._fit_sets: List[PreparedStrings] = []
        self._keys = None
        self._df = None
        self.idf_ = None
        self._vocabulary: Optional[Dict[str, int]] = None
        self._host_regex = None if regex_is_char_class(regex) else regex
        table = np.zeros(128, np.uint8) if self._host_regex is not None else delete_table_for(regex)
        self._params = N.SgVecParams()
        self._params.ngram_size = self.ngram_size
        self._params.ascii_lower = 1 if self.ignore_case else 0
        self._params.dtype = N.np_dtype_code(self.dtype)
        for c in range(128):
            self._params.delete_table[c] = int(table[c])

    @property
    def ctx(self) -> N.Context:
        if self._ctx is None:
            self._ctx = N.default_context()
        return self._ctx

    # ------------------------------------------------------------------ device-level API
    def prepare(self, strings) -> PreparedStrings:
        p = prepare_strings(strings, self.ignore_case, self.normalize_to_ascii, self._host_regex)
        p.dev = self.ctx.strings_from_host(p.data, p.offsets)
        return p

    def fit_prepared(self, sets: Sequence[PreparedStrings]) -> "HipTfidfVectorizer":
        """TfidfVectorizer.fit(concat(sets)) (string_grouper.py:699-707)."""
        self._vocab = self.ctx.vec_fit([s.dev for s in sets], self._params)
        self._fit_sets = list(sets)
        n_terms, n_docs = self.ctx.vocab_size(self._vocab)
        self._keys, self._df = self.ctx.vocab_to_host(self._vocab)
        self.idf_ = idf_from_df(self._df, n_docs, self.dtype)
        self.ctx.vocab_set_idf(self._vocab, self.idf_)
        self._vocabulary = None
        return self

    # ---- the two halves of fit for a caller that shards the strings over several GPUs (distributed.py)
    def fit_begin_prepared(self, sets: Sequence[PreparedStrings]) -> "HipTfidfVectorizer":
        """Tokenise the LOCAL sets and count their document frequencies; ``df_table()`` is then summed across
        ranks and ``fit_end`` finishes the vocabulary + idf identically on every rank."""
        self._vocab = self.ctx.vec_fit_begin([s.dev for s in sets], self._params)
        self._fit_sets = list(sets)
        return self

    def df_table(self):
        """(device pointer, entries, shareable) of the dense int32 document-frequency table of ``fit_begin``."""
        return self.ctx.vocab_df_table(self._vocab)

    def fit_end(self, n_docs_total: int = 0) -> "HipTfidfVectorizer":
        self.ctx.vec_fit_end(self._vocab, n_docs_total)
        n_terms, n_docs = self.ctx.vocab_size(self._vocab)
        self._keys, self._df = self.ctx.vocab_to_host(self._vocab)
        self.idf_ = idf_from_df(self._df, n_docs, self.dtype)
        self.ctx.vocab_set_idf(self._vocab, self.idf_)
        self._vocabulary = None
        return self

    def transform_prepared(self, s: PreparedStrings) -> N.Csr:
        if self._vocab is None:
            raise RuntimeError("vectoriser is not fitted")
        return self.ctx.vec_transform(self._vocab, s.dev)

    # ------------------------------------------------------------------ sklearn-shaped API (seam b1)
    def fit(self, raw_documents, y=None):
        return self.fit_prepared([self.prepare(raw_documents)])

    def transform(self, raw_documents):
        """Always re-reads ``raw_documents`` (as sklearn does): the tokens of fit() are only reused through the
        explicit handles (``prepare`` / ``fit_prepared`` / ``transform_prepared``, or ``fit_transform``)."""
        return self.transform_prepared(self.prepare(raw_documents)).to_scipy()

    def fit_transform(self, raw_documents, y=None):
        p = self.prepare(raw_documents)
        return self.fit_prepared([p]).transform_prepared(p).to_scipy()      # one tokenisation pass

    @property
    def vocabulary_(self) -> Dict[str, int]:
        if self._vocabulary is None:
            if self._keys is None:
                raise AttributeError("vocabulary_ is available after fit()")
            self._vocabulary = {t: i for i, t in enumerate(keys_to_terms(self._keys, self.ngram_size))}
        return self._vocabulary

    def get_feature_names_out(self):
        return np.asarray(keys_to_terms(self._keys, self.ngram_size), dtype=object)
