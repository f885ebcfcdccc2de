"""string_grouper_amd/_hostops.py (libsg_host.so): the threaded object gather and the ASCII column copy return exactly what
the numpy / pyarrow code they stand in for returns -- the same objects, the same bytes -- and step aside for anything else."""
import sys

import numpy as np
import pandas as pd
import pytest

from string_grouper_amd import _hostops as H
from string_grouper_amd import strprep
from string_grouper_amd.synth import synth_names

needs_lib = pytest.mark.skipif(H._load() is None, reason="libsg_host.so not built (python __graft_entry__.py)")


@needs_lib
def test_gather_returns_the_same_objects_and_keeps_the_reference_counts_right():
    names = synth_names(70000, 3)
    names[17] = "".join(["PROBE ", "OBJECT ", "OF ITS OWN"])           # (the generator hands out one object for exact repeats)
    arr = np.array(names, dtype=object)
    rng = np.random.default_rng(1)
    idx = rng.integers(0, len(arr), 200000).astype(np.int64)
    none_before, probe_before = sys.getrefcount(None), sys.getrefcount(names[17])
    got = H.take_objects(arr, idx)
    uses = int((idx == 17).sum())
    after = sys.getrefcount(names[17])       # (measured outside the assert: pytest's rewriting keeps its operands alive)
    assert after == probe_before + uses                                   # one new reference per use
    want = arr.take(idx)
    assert got.dtype == object and got.shape == want.shape
    same = [a is b for a, b in zip(got.tolist(), want.tolist())]
    assert all(same)
    del got, want, same
    after = sys.getrefcount(names[17])
    assert after == probe_before
    none_after = sys.getrefcount(None)
    assert abs(none_after - none_before) < 64                             # the fresh array's references to None were given back
    # other index types, and what numpy refuses
    assert all(a is b for a, b in zip(H.take_objects(arr, idx.astype(np.int32)).tolist(), arr.take(idx).tolist()))
    # ONE object at many positions (a list that repeats a name): still counted exactly -- the counts are raised by atomic adds
    shared = ["".join(["SHARED ", str(i)]) for i in range(40)]
    rep = np.array([shared[i % 40] for i in range(100000)], dtype=object)
    ridx = rng.integers(0, len(rep), 1500000).astype(np.int64)
    base_counts = [sys.getrefcount(x) for x in shared]
    g2 = H.take_objects(rep, ridx)
    now = [sys.getrefcount(x) for x in shared]
    uses40 = np.bincount(ridx % 40, minlength=40)
    assert [a - b for a, b in zip(now, base_counts)] == uses40.tolist()
    del g2
    bad = idx.copy()
    bad[5] = len(arr)
    with pytest.raises(IndexError):
        H.take_objects(arr, bad)


@needs_lib
def test_ascii_columns_are_copied_out_and_everything_else_goes_the_general_way():
    names = synth_names(50000, 5) + ["", "a", "x" * 5000]
    arr = np.array(names, dtype=object)
    data, off = H.ascii_column_bytes(arr)
    assert off[0] == 0 and off[-1] == len(data) == sum(len(s) for s in names)
    assert bytes(data[off[123]:off[124]]).decode() == names[123] and data[off[-2]:].tobytes() == b"x" * 5000
    for odd in ("ÀbracâDABRÀ", 3, None, b"bytes", np.str_("numpy str")):
        mixed = arr.copy()
        mixed[777] = odd
        assert H.ascii_column_bytes(mixed) is None
    # through the column preparation: the same buffers with and without the helper
    s = pd.Series(names)
    d1, o1 = strprep.to_arrow_buffers(s)
    saved, H._lib = H._lib, None
    try:
        d2, o2 = strprep.to_arrow_buffers(s)
    finally:
        H._lib = saved
    assert np.array_equal(d1, d2) and np.array_equal(o1, o2)
    uni = pd.Series(names[:5000] + ["Ünïcödé GmbH"])
    d3, o3 = strprep.to_arrow_buffers(uni)
    assert bytes(d3[o3[-2]:]).decode("utf-8") == "Ünïcödé GmbH"
    with pytest.raises(TypeError):
        strprep.to_arrow_buffers(pd.Series(names[:5000] + [None], dtype=object))


@needs_lib
def test_expansions_of_the_match_list_equal_numpys():
    rng = np.random.default_rng(3)
    counts = rng.integers(0, 9, 120000)
    counts[[0, 5, 119999]] = 0                                             # empty rows at both ends
    row_ptr = np.zeros(len(counts) + 1, np.int64)
    np.cumsum(counts, out=row_ptr[1:])
    assert row_ptr[-1] > 300000
    want = np.repeat(np.arange(len(counts), dtype=np.int64), counts)
    got = H.expand_rows(row_ptr)
    assert got.dtype == np.int64 and np.array_equal(got, want)
    assert np.array_equal(H.expand_rows(row_ptr[:50]), np.repeat(np.arange(49, dtype=np.int64), counts[:49]))   # small: numpy's
    cols = rng.integers(-5, 2 ** 31 - 1, 400000).astype(np.int32)
    w = H.widen(cols, np.int64)
    assert w.dtype == np.int64 and np.array_equal(w, cols.astype(np.int64))
    vals = rng.random(400000).astype(np.float32)
    v = H.widen(vals, np.float64)
    assert v.dtype == np.float64 and np.array_equal(v, vals.astype(np.float64))
    assert H.widen(vals[::2], np.float64).dtype == np.float64             # not contiguous: numpy's
    assert np.array_equal(H.widen(cols, np.float64), cols.astype(np.float64))      # another pair of types: numpy's
    pos = rng.integers(0, 10 ** 6, 500000).astype(np.int64)
    a = H.affine_i64(pos, 0, 1)
    assert a is not pos and np.array_equal(a, pos)
    assert np.array_equal(H.affine_i64(pos, 7, 3), 7 + pos * 3)
    assert np.array_equal(H.affine_i64(pos[:100], 7, 3), 7 + pos[:100] * 3)
