"""Host-side string preparation for the device tokeniser (K1): the part of the reference's analyzer
``StringGrouper.n_grams`` (string_grouper/string_grouper.py:365-378)

    lower() -> NFKD + encode('ASCII', 'ignore') -> re.sub(regex, '')

that has no byte-level definition.  Pure-ASCII columns with a character-class regex (the default) need none of it:
the device lower-cases, drops and n-grams the bytes as they are.  Everything else is prepared here, VECTORISED
(numpy over code-point arrays; Python's own ``str.lower`` / ``unicodedata.normalize`` / ``re`` are consulted once
per DISTINCT code point, so the semantics are the reference's by construction):

* ``normalize_to_ascii=True`` (default): the ASCII part of NFKD(lower(s)) is the concatenation over the code points
  c of s of the ASCII part of NFKD(lower(c)) -- canonical reordering only permutes combining marks, which are not
  ASCII; lower() is context free except for the final-sigma rule, whose two outcomes are both non-ASCII.  So every
  non-ASCII code point is replaced by a (possibly empty) ASCII string from a table, e.g. 'é'->'e', 'ß'->'ss',
  '™'->'TM' (NOT lower-cased: the reference lower-cases before it decomposes), 'ﬃ'->'ffi', '東'->''.
  The result is an ASCII byte column; the regex is still applied on the device (delete table).
* ``normalize_to_ascii=False`` with non-ASCII characters: the n-grams are over code points.  The column becomes a
  SYMBOL column: uint32 code points, lower-cased (per code point; rows containing 'Σ' go through ``str.lower`` because
  of the final-sigma rule), regex-deleted (per code point), ready to be ranked into the fit's alphabet.
* a regex that is not a character class (its matches span several characters or depend on context): ``re.sub`` per
  string, as the reference does -- there is no vectorised equivalent of Python's regex semantics.

Whenever the host has lower-cased a column the device must not do it again (``prelowered``): 'TM' from '™' stays.
"""
from __future__ import annotations

import re
from typing import Optional
from unicodedata import normalize as _ucd_normalize

import numpy as np

try:                                   # Python >= 3.11
    import re._parser as _sre_parse    # type: ignore
    import re._constants as _sre_c     # type: ignore
except ImportError:                    # Python 3.10
    import sre_parse as _sre_parse     # type: ignore
    import sre_constants as _sre_c     # type: ignore

SIGMA = 0x03A3                         # GREEK CAPITAL LETTER SIGMA: the one context-sensitive case of str.lower()


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


class StringColumn:
    """A string column ready for the device.

    kind "bytes":   ``data`` uint8 (UTF-8 / ASCII) + ``offsets`` int64[n + 1]; the device drops bytes >= 0x80, lower-cases
                    A-Z unless ``prelowered``, deletes the bytes of the regex's delete table.
    kind "symbols": ``data`` uint32 code points, already lower-cased and regex-deleted + ``offsets`` int64[n + 1] in
                    symbols; the vectoriser ranks them into the alphabet of the fit and uploads uint16 ranks."""

    def __init__(self, kind: str, data: np.ndarray, offsets: np.ndarray, prelowered: bool = False):
        self.kind = kind
        self.data = data
        self.offsets = offsets
        self.prelowered = prelowered
        self.n = len(offsets) - 1
        self.dev = None            # device handle (N.Strings), set by the vectoriser


# ------------------------------------------------------------------------------------------------ Arrow buffers
def to_arrow_buffers(strings):
    """(UTF-8 bytes uint8, offsets int64[n + 1]) of a sequence of str / pandas Series, without a per-string loop."""
    import pyarrow as pa
    if hasattr(strings, "array") and hasattr(strings.array, "_pa_array"):       # pandas ArrowExtensionArray: no copy
        arr = strings.array._pa_array.combine_chunks()
        if not pa.types.is_large_string(arr.type):
            arr = arr.cast(pa.large_string())
    else:
        values = strings.to_numpy() if hasattr(strings, "to_numpy") else np.asarray(strings, dtype=object)
        # (round 6: a column of ASCII str objects -- every list of company names -- is copied out by a few host threads,
        #  _hostops.ascii_column_bytes; anything else in it and pyarrow looks at the column: 18 -> ms at 663 k names)
        if isinstance(values, np.ndarray) and values.dtype == object:
            from . import _hostops
            fast = _hostops.ascii_column_bytes(values)
            if fast is not None:
                return fast
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


def _ascii_lower(data: np.ndarray) -> np.ndarray:
    up = (data >= 65) & (data <= 90)
    if not up.any():
        return data
    out = data.copy()
    out[up] += 32
    return out


def _decode_rows(data: np.ndarray, offsets: np.ndarray, rows: np.ndarray):
    """Code points of the given rows: (uint32 code points of all those rows back to back, int64 offsets per row)."""
    starts, ends = offsets[rows], offsets[rows + 1]
    if len(rows) == len(offsets) - 1:
        raw = data
    else:
        raw = np.concatenate([data[s:e] for s, e in zip(starts.tolist(), ends.tolist())]) if len(rows) else data[:0]
    is_start = (raw & 0xC0) != 0x80                         # not a UTF-8 continuation byte
    csum = np.concatenate([[0], np.cumsum(is_start, dtype=np.int64)])
    lens = ends - starts
    bpos = np.concatenate([[0], np.cumsum(lens)])
    cps = np.frombuffer(raw.tobytes().decode("utf-8").encode("utf-32-le"), dtype=np.uint32)
    return cps, csum[bpos]


def _expand(cps: np.ndarray, char_off: np.ndarray, mapping):
    """Replace every code point c by the sequence ``mapping(c)`` (a tuple of ints, possibly empty): vectorised over the
    positions, ``mapping`` is called once per distinct code point.  Returns (new code points, new row offsets)."""
    if cps.size == 0:
        return cps, char_off
    uniq, inv = np.unique(cps, return_inverse=True)
    seqs = [mapping(int(c)) for c in uniq.tolist()]
    lens_u = np.array([len(s) for s in seqs], dtype=np.int64)
    width = int(lens_u.max()) if len(seqs) else 0
    table = np.zeros((len(seqs), max(width, 1)), dtype=np.uint32)
    for i, s in enumerate(seqs):
        table[i, :len(s)] = s
    out_len = lens_u[inv]
    csum = np.concatenate([[0], np.cumsum(out_len)])
    out = np.empty(int(csum[-1]), dtype=np.uint32)
    starts = csum[:-1]
    for j in range(width):
        m = out_len > j
        out[starts[m] + j] = table[inv[m], j]
    return out, csum[char_off]


def _splice(data, offsets, rows, new_bytes, new_off):
    """The byte column with the given rows replaced by new_bytes[new_off[i]:new_off[i + 1]]."""
    n = len(offsets) - 1
    old_len = np.diff(offsets)
    new_len = old_len.copy()
    new_len[rows] = np.diff(new_off)
    out_off = np.concatenate([[0], np.cumsum(new_len)]).astype(np.int64)
    out = np.empty(int(out_off[-1]), dtype=np.uint8)
    if len(rows) <= 4096:                                   # few rows: copy the untouched runs between them
        prev = 0
        for i, r in enumerate(rows.tolist()):
            if r > prev:
                out[out_off[prev]:out_off[r]] = data[offsets[prev]:offsets[r]]
            out[out_off[r]:out_off[r + 1]] = new_bytes[new_off[i]:new_off[i + 1]]
            prev = r + 1
        if prev < n:
            out[out_off[prev]:] = data[offsets[prev]:]
    else:                                                   # many rows: one gather for the untouched bytes
        touched = np.zeros(n, dtype=bool)
        touched[rows] = True
        keep_rows = np.flatnonzero(~touched)
        shift = np.repeat(out_off[keep_rows] - offsets[keep_rows], old_len[keep_rows])
        src = np.flatnonzero(np.repeat(~touched, old_len))
        out[src + shift] = data[src]
        dst = np.repeat(out_off[rows] - new_off[:-1], np.diff(new_off)) + np.arange(len(new_bytes))
        out[dst] = new_bytes
    return out, out_off


# ------------------------------------------------------------------------------------------------ the analyzer prefix
def prepare_column(strings, ignore_case: bool, normalize_to_ascii: bool, regex: str) -> StringColumn:
    """The host step of n_grams (string_grouper.py:372-376) for one string column."""
    data, offsets = to_arrow_buffers(strings)
    n = len(offsets) - 1
    char_class = regex_is_char_class(regex)
    if not char_class:
        # re.sub with Python's semantics, per string, on the lower-cased / normalised text -- as the reference does
        rx = re.compile(regex)
        values = strings.tolist() if hasattr(strings, "tolist") else list(strings)
        if ignore_case:
            values = [s.lower() for s in values]
        if normalize_to_ascii:
            values = [_ucd_normalize('NFKD', s).encode('ASCII', 'ignore').decode() for s in values]
        values = [rx.sub('', s) for s in values]
        data, offsets = to_arrow_buffers(np.asarray(values, dtype=object) if values else np.zeros(0, dtype=object))
        if data.size == 0 or int(data.max()) < 0x80:
            return StringColumn("bytes", data, offsets, prelowered=True)          # nothing left for the device to delete
        cps, coff = _decode_rows(data, offsets, np.arange(n))
        return StringColumn("symbols", cps, coff, prelowered=True)

    if data.size == 0 or int(data.max()) < 0x80:
        return StringColumn("bytes", data, offsets)                                # the device does all of it

    hi = np.flatnonzero(data >= 0x80)
    rows = np.unique(np.searchsorted(offsets, hi, side="right") - 1)               # rows with a non-ASCII character

    def lower_of(c: int):
        return chr(c).lower() if ignore_case else chr(c)

    if normalize_to_ascii:
        cps, coff = _decode_rows(data, offsets, rows)

        def ascii_part(c: int):
            if c < 128:
                return (ord(lower_of(c)),)
            return tuple(ord(x) for x in _ucd_normalize('NFKD', lower_of(c)) if ord(x) < 128)
        new_cps, new_off = _expand(cps, coff, ascii_part)
        out, out_off = _splice(_ascii_lower(data) if ignore_case else data, offsets, rows, new_cps.astype(np.uint8), new_off)
        return StringColumn("bytes", out, out_off, prelowered=ignore_case)

    # ---- code-point n-grams: the whole column as symbols
    rx = re.compile(regex)
    cps, coff = _decode_rows(data, offsets, np.arange(n))
    if ignore_case and (cps == SIGMA).any():
        # final-sigma rule: rows with a capital sigma through str.lower(); everything else per code point
        values = strings.tolist() if hasattr(strings, "tolist") else list(strings)
        srows = np.unique(np.searchsorted(coff, np.flatnonzero(cps == SIGMA), side="right") - 1)
        lowered = [values[r].lower() for r in srows.tolist()]
        ldata, loff = to_arrow_buffers(np.asarray(lowered, dtype=object))
        lcps, lcoff = _decode_rows(ldata, loff, np.arange(len(lowered)))
        cps, coff = _splice_cps(cps, coff, srows, lcps, lcoff)

    def lowered_and_kept(c: int):
        s = lower_of(c) if c != SIGMA else chr(c)            # (a sigma left here was lower-cased with its row)
        return tuple(ord(x) for x in s if not rx.fullmatch(x))
    new_cps, new_off = _expand(cps, coff, lowered_and_kept)
    return StringColumn("symbols", new_cps, new_off, prelowered=True)


def _splice_cps(cps, coff, rows, new_cps, new_off):
    n = len(coff) - 1
    old_len = np.diff(coff)
    new_len = old_len.copy()
    new_len[rows] = np.diff(new_off)
    out_off = np.concatenate([[0], np.cumsum(new_len)]).astype(np.int64)
    out = np.empty(int(out_off[-1]), dtype=np.uint32)
    prev = 0
    for i, r in enumerate(rows.tolist()):
        if r > prev:
            out[out_off[prev]:out_off[r]] = cps[coff[prev]:coff[r]]
        out[out_off[r]:out_off[r + 1]] = new_cps[new_off[i]:new_off[i + 1]]
        prev = r + 1
    if prev < n:
        out[out_off[prev]:] = cps[coff[prev]:]
    return out, out_off


def bytes_column_to_symbols(col: StringColumn, ignore_case: bool, delete_table: np.ndarray) -> StringColumn:
    """A byte column as the symbol column the device would see (lower, drop >= 0x80, delete): needed when another column
    of the same fit carries non-ASCII symbols and all columns must share one alphabet."""
    data = col.data
    if ignore_case and not col.prelowered:
        data = _ascii_lower(data)
    keep = (data < 0x80)
    keep &= delete_table[np.minimum(data, 127)] == 0
    csum = np.concatenate([[0], np.cumsum(keep, dtype=np.int64)])
    return StringColumn("symbols", data[keep].astype(np.uint32), csum[col.offsets], prelowered=True)
