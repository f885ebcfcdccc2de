"""The pruning rule of the pruned multiply (string_grouper_amd/csrc/sg_spgemm_pruned.hip), restated in
numpy with the kernel's arithmetic (float32 bound, upward-quantised filter postings, 2^15 fixed point in
24-bit integer multiplies, the same slack terms)
and checked against the oracle: every pair the oracle keeps must be among the kernel's survivors.
This pins the MATH of the filter on the CPU; the kernel itself is compared bit for bit with the
oracle by the GPU parity tests."""
import numpy as np
import pytest

from oracle import oracle as O
from string_grouper_amd.synth import synth_names

f32 = np.float32
AB = 13                      # address + half bits of a filter posting at tile_log2 = 12
BQ_MAX = (1 << (24 - AB)) - 1


def quantise_right(m, mt, freq_min, norm_up):
    """K3's filter postings: per right row fq (8 bits, frequent-part norm) and per posting bq (11 bits at the
    default tile of 4096 columns), both rounded up relative to norm_up (sg_postings.hip, emit_posting)."""
    df = np.diff(mt.indptr)
    inv = f32(1.0) / f32(norm_up)
    frequent = df[m.indices] >= freq_min
    f2 = np.zeros(m.shape[0])
    np.add.at(f2, np.repeat(np.arange(m.shape[0]), np.diff(m.indptr))[frequent],
              m.data[frequent].astype(np.float64) ** 2)
    fq = np.minimum(255, np.ceil(np.nextafter(np.sqrt(f2).astype(f32), f32(2)) * inv * f32(255.0) * f32(1.000002)))
    bq_t = np.minimum(BQ_MAX, np.ceil(mt.data.astype(f32) * inv * f32(BQ_MAX) * f32(1.000002)))
    return fq.astype(np.int64), bq_t.astype(np.int64)


def survivors_of_row(a_idx, a_val, Bt_indptr, Bt_rows, bq_t, fq, thr, delta, norm_up, freq_min):
    """Columns the kernel would score exactly for one left row (model of the kernel, float32 data)."""
    nnz = len(a_idx)
    df = (Bt_indptr[a_idx + 1] - Bt_indptr[a_idx]).astype(np.int64)
    beta = thr - delta
    budget = np.nextafter(f32((beta / norm_up) ** 2 * (1.0 - 1e-6)), f32(0))
    w = (a_val.astype(f32) * a_val.astype(f32) * f32(1.00001)).astype(f32)
    cum = np.zeros(nnz, f32)
    for lane in range(nnz):
        c = f32(0)
        for q in range(nnz):
            if df[q] > df[lane] or (df[q] == df[lane] and q <= lane):
                c = f32(c + w[q])
        cum[lane] = c
    in_s = (cum <= budget) & (df >= freq_min)
    in_p = ~in_s
    if not in_p.any():
        return np.zeros(0, np.int64), 0
    bs2 = cum[in_s].max() if in_s.any() else f32(0)
    b_s = f32(f32(np.sqrt(bs2)) * f32(1.000002))
    t0 = f32(f32(f32(thr) - f32(1e-5)) * f32(32768.0)) - f32(2.0)
    c1 = f32(f32(f32(b_s * f32(norm_up)) * f32(32768.0 / 255.0)) * f32(1.000002))
    n_p = int(in_p.sum())
    T0 = int(np.floor(f32(t0 * f32(256.0)))) - 256 * n_p
    C1 = int(f32(c1 * f32(256.0))) + 1
    assert T0 - C1 * 255 >= 256
    q = {}
    streamed = 0
    for t in np.nonzero(in_p)[0]:
        lo, hi = Bt_indptr[a_idx[t]], Bt_indptr[a_idx[t] + 1]
        c_a = f32(f32(f32(f32(a_val[t]) * f32(norm_up)) * f32(32768.0 / BQ_MAX)) * f32(1.000002))
        CA = int(f32(c_a * f32(1 << (32 - AB)))) + 1
        assert CA < (1 << 24)
        x = (CA * (bq_t[lo:hi].astype(np.int64) << AB)) >> 32      # v_mul_hi_u32_u24
        streamed += hi - lo
        for j, xv in zip(Bt_rows[lo:hi], x):
            q[j] = q.get(j, 0) + int(xv)
    assert max(q.values()) < 65536
    surv = np.array(sorted(j for j, v in q.items() if v >= ((T0 - C1 * int(fq[j])) >> 8)), dtype=np.int64)
    return surv, streamed


@pytest.mark.parametrize("thr,delta,freq", [(0.8, 0.2, 0.003), (0.8, 0.05, 0.0), (0.5, 0.2, 0.01), (0.95, 0.3, 0.05),
                                            (0.8, 0.1, 0.003)])
def test_survivors_cover_every_oracle_match(thr, delta, freq):
    names = synth_names(3000, 77)
    (m,), _, _ = O.tfidf_sklearn(names, [names], dtype=np.float32)
    m = m.tocsr()
    m.sort_indices()
    mt = m.T.tocsr()
    mt.sort_indices()
    C = O.sp_matmul_topn(m, m.T.tocsr(), 10_000, thr, sort=True)   # uncapped: every pair above thr
    norm_up = np.nextafter(f32(np.sqrt(f32(np.asarray(m.multiply(m).sum(axis=1)).max())) * f32(1.000001)), f32(2))
    freq_min = max(1, int(freq * m.shape[0]))
    fq, bq_t = quantise_right(m, mt, freq_min, norm_up)
    total_streamed = total_full = total_surv = 0
    for i in range(0, m.shape[0], 7):
        lo, hi = m.indptr[i], m.indptr[i + 1]
        surv, streamed = survivors_of_row(m.indices[lo:hi], m.data[lo:hi], mt.indptr, mt.indices, bq_t, fq, thr, delta,
                                          norm_up, freq_min)
        want = C.indices[C.indptr[i]:C.indptr[i + 1]]
        assert set(want) <= set(surv), (i, sorted(set(want) - set(surv)))
        total_streamed += streamed
        total_surv += len(surv)
        total_full += int((mt.indptr[m.indices[lo:hi] + 1] - mt.indptr[m.indices[lo:hi]]).sum())
    assert total_streamed < total_full   # the filter does prune


# ---------------------------------------------------------------------------------------------------------------------
# Stream form (round 3): eight tiles of 4096 columns share ONE accumulator tile (column c -> accumulator c mod 4096),
# bq has 8 bits, the fixed point is 2^12, and a posting records its column when `old + x >= tq_j` -- whatever the
# other columns folded onto the accumulator have added before.  The model runs the accumulation in a RANDOM order of
# the postings (the kernel's order depends on how the lanes are dealt) and must still record every oracle match.
#
# Round 4: the posting's fields are cut for the multiply's instructions (sg_postings.hip, emit_posting).  Its low 24
# bits go into v_mul_hi_u32_u24 AS ONE NUMBER (K3 rounds bq so that {bq, fold, word, half} >= v / norm_up * 255 * 2^16),
# its upper 16 bits into v_mad_i32_i16 as F - 32768 (K3 rounds fq so that F = {fq, bq} >= f / norm_up * 255 * 256), and the
# test is ((old + x) << 16) >= T16 - C16 * F.  `eight_bit=True` is round 3's arithmetic (masked bq, 8-bit fq), kept to
# check that the new bounds are as tight.
FOLD_LOG2 = 3
FB = AB + FOLD_LOG2
BQ_MAX_S = (1 << (24 - FB)) - 1
SCALE_S = 32768 >> FOLD_LOG2
F16_MAX = 255 * 256


def quantise_right_stream(m, mt, freq_min, norm_up, tile=4096):
    """-> (fq8 per right row, bq8 per posting) of round 3, and the 32-bit filter postings of round 4 (mt's order)."""
    df = np.diff(mt.indptr)
    inv = f32(1.0) / f32(norm_up)
    frequent = df[m.indices] >= freq_min
    f2 = np.zeros(m.shape[0])
    np.add.at(f2, np.repeat(np.arange(m.shape[0]), np.diff(m.indptr))[frequent],
              m.data[frequent].astype(np.float64) ** 2)
    fr = (np.nextafter(np.sqrt(f2).astype(f32), f32(2)) * inv).astype(f32)       # frequent_norm_ratio
    fq8 = np.minimum(255, np.ceil(fr * f32(255.0) * f32(1.000002))).astype(np.int64)
    v = mt.data.astype(f32)
    bq8 = np.minimum(BQ_MAX_S, np.ceil(v * inv * f32(BQ_MAX_S) * f32(1.000002))).astype(np.int64)
    j = mt.indices.astype(np.int64)
    c, fold = j % tile, (j // tile) % (1 << FOLD_LOG2)
    low = ((c >> 1) << 2) | ((c & 1) << 1) | (fold << AB)
    b24 = np.minimum(255 << 16, np.ceil(f32(v * inv) * f32(255.0 * 65536.0) * f32(1.000002))).astype(np.int64)
    bq = np.where(b24 > low, (b24 - low + 65535) >> 16, 0)
    f16 = np.minimum(F16_MAX, np.ceil(fr[j] * f32(F16_MAX) * f32(1.000002))).astype(np.int64)
    fq = np.where(f16 > bq, (f16 - bq + 255) >> 8, 0)
    assert bq.max() <= 255 and fq.max() <= 255
    postings = low | (bq << 16) | ((fq ^ 0x80) << 24)
    # what the bounds must hold (the kernel's two numbers are upper bounds of the exact ratios)
    assert ((postings & 0xFFFFFF) >= np.ceil(v.astype(np.float64) / float(norm_up) * 255 * 65536 - 1e-3)).all()
    F = ((postings >> 16) ^ 0x8000)
    assert (F >= np.minimum(F16_MAX, np.ceil(np.sqrt(f2)[j] / float(norm_up) * F16_MAX - 1e-3))).all()
    return fq8, bq8, postings


def records_of_row_stream(a_idx, a_val, Bt_indptr, Bt_rows, bq_t, fq, postings, thr, delta, norm_up, freq_min, rng, tile=4096,
                          eight_bit=False):
    """Columns the stream form records for one left row, repeats and false positives included."""
    nnz = len(a_idx)
    df = (Bt_indptr[a_idx + 1] - Bt_indptr[a_idx]).astype(np.int64)
    beta = thr - delta
    budget = np.nextafter(f32((beta / norm_up) ** 2 * (1.0 - 1e-6)), f32(0))
    w = (a_val.astype(f32) * a_val.astype(f32) * f32(1.00001)).astype(f32)
    cum = np.zeros(nnz, f32)
    for lane in range(nnz):
        c = f32(0)
        for q in range(nnz):
            if df[q] > df[lane] or (df[q] == df[lane] and q <= lane):
                c = f32(c + w[q])
        cum[lane] = c
    in_s = (cum <= budget) & (df >= freq_min)
    in_p = ~in_s
    if not in_p.any():
        return np.zeros(0, np.int64), 0
    bs2 = cum[in_s].max() if in_s.any() else f32(0)
    b_s = f32(f32(np.sqrt(bs2)) * f32(1.000002))
    t0 = f32(f32(f32(thr) - f32(1e-5)) * f32(SCALE_S)) - f32(2.0)
    c1 = f32(f32(f32(b_s * f32(norm_up)) * f32(SCALE_S / 255.0)) * f32(1.000002))
    n_p = int(in_p.sum())
    T0 = int(np.floor(f32(t0 * f32(256.0)))) - 256 * n_p
    C1 = int(f32(c1 * f32(256.0))) + 1
    C16 = int(f32(f32(f32(b_s * f32(norm_up)) * f32(f32(SCALE_S) * f32(65536.0) / f32(65280.0))) * f32(1.000002))) + 1
    T16 = T0 << 8
    if (T0 - C1 * 255 < 256) if eight_bit else (T16 - C16 * 65535 < 65536):
        return None, 0          # the kernel hands such a row to the exact kernel
    assert C16 < (1 << 15) and T16 < (1 << 31)
    T0s, C1n = T16 - 32768 * C16, -C16
    cols, xs, rs = [], [], []
    for t in np.nonzero(in_p)[0]:
        lo, hi = Bt_indptr[a_idx[t]], Bt_indptr[a_idx[t] + 1]
        c_a = f32(f32(f32(f32(a_val[t]) * f32(norm_up)) * f32(SCALE_S / BQ_MAX_S)) * f32(1.000002))
        CA = int(f32(c_a * f32(1 << (32 - FB)))) + 1
        assert CA < (1 << 24)
        if eight_bit:
            xs.append((CA * (bq_t[lo:hi].astype(np.int64) << FB)) >> 32)      # v_mul_hi_u32_u24 of the masked field
        else:
            xs.append((CA * (postings[lo:hi] & 0xFFFFFF)) >> 32)              # v_mul_hi_u32_u24 of the posting as it is
        cols.append(Bt_rows[lo:hi])
        rs.append(postings[lo:hi])
    cols = np.concatenate(cols)
    xs = np.concatenate(xs)
    rs = np.concatenate(rs)
    order = rng.permutation(len(cols))
    acc = {}
    rec = []
    for j, x, r in zip(cols[order], xs[order], rs[order]):
        key = (int(j) // (tile << FOLD_LOG2), int(j) % tile)     # (visit, accumulator)
        old = acc.get(key, 0)
        acc[key] = old + int(x)
        assert acc[key] < 65536                                  # sixteen bits hold the sums of all folded columns
        if eight_bit:
            fired = old + int(x) >= ((T0 - C1 * int(fq[j])) >> 8)
        else:
            hi16 = int(r) >> 16
            d = T0s + C1n * (hi16 - 65536 if hi16 >= 32768 else hi16)   # v_mad_i32_i16 on the posting's upper half
            assert d >= 65536
            fired = ((old + int(x)) << 16) >= d
            col_bits = (int(r) >> 1) & 0x7FFF                           # v_bfe_u32 r, 1, 15: the column inside its super-tile
            assert col_bits == ((int(j) // tile) % (1 << FOLD_LOG2)) * 4096 + int(j) % tile
        if fired:
            rec.append(int(j))
    return np.array(rec, dtype=np.int64), len(cols)


@pytest.mark.parametrize("thr,delta,freq,tile", [(0.8, 0.05, 0.005, 4096), (0.8, 0.05, 0.0045, 16), (0.5, 0.2, 0.01, 16),
                                                 (0.95, 0.3, 0.05, 64), (0.6, 0.02, 0.0, 16), (0.8, 0.03, 0.005, 4096)])
def test_stream_form_records_cover_every_oracle_match(thr, delta, freq, tile):
    """`tile` much smaller than the kernel's 4096 folds 3000 columns as heavily as 663 k columns fold in the kernel."""
    names = synth_names(3000, 78)
    (m,), _, _ = O.tfidf_sklearn(names, [names], dtype=np.float32)
    m = m.tocsr()
    m.sort_indices()
    mt = m.T.tocsr()
    mt.sort_indices()
    C = O.sp_matmul_topn(m, m.T.tocsr(), 10_000, thr, sort=True)
    norm_up = np.nextafter(f32(np.sqrt(f32(np.asarray(m.multiply(m).sum(axis=1)).max())) * f32(1.000001)), f32(2))
    freq_min = max(1, int(freq * m.shape[0]))
    fq, bq_t, postings = quantise_right_stream(m, mt, freq_min, norm_up, tile=tile)
    n_true = 0
    n_rec = {False: 0, True: 0}
    for eight_bit in (False, True):
        rng = np.random.default_rng(5)
        for i in range(0, m.shape[0], 7):
            lo, hi = m.indptr[i], m.indptr[i + 1]
            rec, _ = records_of_row_stream(m.indices[lo:hi], m.data[lo:hi], mt.indptr, mt.indices, bq_t, fq, postings, thr,
                                           delta, norm_up, freq_min, rng, tile=tile, eight_bit=eight_bit)
            if rec is None:
                continue
            want = C.indices[C.indptr[i]:C.indptr[i + 1]]
            assert set(want) <= set(rec.tolist()), (eight_bit, i, sorted(set(want) - set(rec.tolist())))
            n_rec[eight_bit] += len(set(rec.tolist()))
            n_true += len(want) if not eight_bit else 0
    assert n_true > 0 and n_rec[False] >= n_true
    # as tight as round 3's bounds (the sixteen-bit F is tighter than an 8-bit fq; the value pays for the address bits)
    assert n_rec[False] <= 1.03 * n_rec[True] + 5, n_rec


def test_stream_form_posting_of_a_full_value_in_a_tile_s_first_columns():
    """A row of ONE term has the value 1 = norm_up; the safety factor of the quantisation lifts it a few units above
    255 * 2^16, and in a column whose address bits are smaller than that excess bq came out as 256 -- the field wrapped
    to 0 and carried into fq (round 4, caught by the GPU's seeded random jobs: 'ADI' / 'ADI.' did not find themselves)."""
    import scipy.sparse as sp
    n = 40
    rows = np.arange(n)
    m = sp.csr_matrix((np.ones(n, np.float32), (rows, np.zeros(n, np.int64))), shape=(n, 3))   # every row: term 0, value 1
    mt = m.T.tocsr()
    norm_up = np.nextafter(f32(1.0) * f32(1.000001), f32(2))
    fq8, bq8, postings = quantise_right_stream(m, mt, 1 << 30, norm_up)
    bq = (postings >> 16) & 0xFF
    assert (bq == 255).all(), bq
    assert (((postings >> 24) ^ 0x80) == 0).all()          # no frequent part, and nothing carried into the field


# ---------------------------------------------------------------------------------------------------------------------
# Round 5: the SECOND filter -- an 8-bit copy of every right-hand row (sg_postings.hip: q8_write_unit) and the bound the
# multiply computes from it before it scores a candidate exactly (sg_spgemm_pruned.hip: q8_passes).  Restated with the
# kernels' float32 operations, in their order; the claim under test: a pair whose EXACT score (the reference's
# arithmetic, in the matrix dtype) is above the threshold always passes.
Q8_MAX_ENTRIES = 60


def q8_quantise(vals, norm_up):
    """bq of K3: ceil(float(v) * (1 / norm_up) * 255 * 1.000002), cut to [1, 255] -- every product a float32 operation."""
    inv = f32(1.0) / f32(norm_up)
    q = np.ceil(((vals.astype(f32) * inv).astype(f32) * f32(255.0)).astype(f32) * f32(1.000002))
    return np.clip(q, 1, 255).astype(np.int64)


def q8_bar(thr, norm_up):
    """((float)thr - 3e-5f) * q8_scale, q8_scale = 255 / norm_up * (1 - 1e-6) rounded down (sg_postings.hip)."""
    scale = np.nextafter(f32(255.0 / float(norm_up) * (1.0 - 1e-6)), f32(0))
    return f32(f32(f32(thr) - f32(3e-5)) * scale)


def q8_bound(a_idx, a_val, b_idx, bq):
    """U of the kernel: over the candidate's entries in ascending term order, ub = fma(float(a_k), float(bq_k), ub)."""
    a = dict(zip(a_idx.tolist(), a_val.astype(f32).tolist()))
    ub = f32(0)
    for k, q in zip(b_idx.tolist(), bq.tolist()):
        ub = f32(np.float64(a.get(k, 0.0)) * np.float64(q) + np.float64(ub))      # (exact in float64, one rounding: an fma)
    return ub


def exact_score(a_idx, a_val, b_idx, b_val, dtype):
    """The reference's arithmetic: ascending k over row j, product and sum rounded separately, in the matrix dtype."""
    a = dict(zip(a_idx.tolist(), a_val.tolist()))
    s = dtype(0)
    for k, v in zip(b_idx.tolist(), b_val.tolist()):
        s = dtype(s + dtype(dtype(a.get(k, 0.0)) * dtype(v)))
    return s


def q8_passes(m, i, j, thr, norm_up):
    bi, bv = m.indices[m.indptr[j]:m.indptr[j + 1]], m.data[m.indptr[j]:m.indptr[j + 1]]
    if len(bi) > Q8_MAX_ENTRIES:
        return True                      # no 8-bit copy: always scored
    ai, av = m.indices[m.indptr[i]:m.indptr[i + 1]], m.data[m.indptr[i]:m.indptr[i + 1]]
    return bool(q8_bound(ai, av, bi, q8_quantise(bv, norm_up)) >= q8_bar(thr, norm_up))


def _norm_up(m):
    return np.nextafter(f32(np.sqrt(f32(np.asarray(m.multiply(m).sum(axis=1)).max())) * f32(1.000001)), f32(2))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("thr", [0.5, 0.8, 0.95])
def test_second_filter_passes_every_pair_above_the_threshold(dtype, thr):
    """Every match the oracle keeps -- hubs of identical and near-identical names included -- passes the 8-bit bound, and
    the bound rejects most of what is far below the threshold (it is a filter, not a formality)."""
    names = list(synth_names(4000, 321))
    for k in range(60):                                   # a hub: identical rows and one-character variants
        names[50 * k] = "NORTHERN LIGHTS HOLDING CO" + ("" if k % 3 else " " + "ABC"[k % 3])
    (m,), _, _ = O.tfidf_sklearn(names, [names], dtype=dtype)
    m = m.tocsr()
    m.sort_indices()
    norm_up = _norm_up(m)
    C = O.sp_matmul_topn(m, m.T.tocsr(), 100000, thr, sort=True)
    rng = np.random.default_rng(7)
    rows = np.unique(np.concatenate([rng.integers(0, m.shape[0], 500), np.arange(0, 3000, 50)]))
    kept = 0
    for i in rows:
        for j in C.indices[C.indptr[i]:C.indptr[i + 1]]:
            assert q8_passes(m, i, j, dtype(thr), norm_up), (i, j)
            kept += 1
    assert kept > len(rows)
    rejected = sum(not q8_passes(m, i, j, dtype(thr), norm_up) for i in rows[:200] for j in rng.integers(0, m.shape[0], 20))
    assert rejected > 0.9 * 200 * 20


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_second_filter_at_the_threshold_plus_and_minus_one_ulp(dtype):
    """The tightest cases: the threshold set ONE ULP under a pair's exact score (the pair is a match and must pass), for
    pairs whose 8-bit values are exact (value / norm_up * 255 is an integer up to rounding: the bound has no slack but
    the 3e-5 allowance) and for rows of the copy's full 60 entries; one ulp ABOVE the score the pair is no match, and
    whether it passes does not matter.  Also a full-length left row (128 terms): the float sum loses the most there."""
    rng = np.random.default_rng(11)
    n_terms = 5000
    checked = 0
    for trial in range(400):
        n_b = int(rng.choice([3, 8, 19, 40, 60]))
        n_a = int(rng.choice([n_b, 64, 128]))
        b_idx = np.sort(rng.choice(n_terms, n_b, replace=False))
        extra = np.setdiff1d(rng.choice(n_terms, n_a, replace=False), b_idx)[: max(n_a - n_b, 0)]
        a_idx = np.sort(np.concatenate([b_idx, extra]))
        if trial % 2:                                     # 8-bit exact values: multiples of norm_up / 255
            q = rng.integers(1, 40, n_b).astype(np.float64)
            b_val = q / np.sqrt((q * q).sum())
            norm_up = np.nextafter(f32(f32(1.0) * f32(1.000001)), f32(2))
            b_val = (np.round(b_val * 255 / float(norm_up)) * float(norm_up) / 255.0)
            b_val = b_val[b_val > 0] if (b_val > 0).all() else np.maximum(b_val, float(norm_up) / 255.0)
        else:
            b_val = rng.random(n_b) + 0.05
            b_val /= np.sqrt((b_val * b_val).sum())
            norm_up = np.nextafter(f32(f32(1.0) * f32(1.000001)), f32(2))
        b_val = b_val.astype(dtype)
        a_val = np.zeros(len(a_idx))
        pos = np.searchsorted(a_idx, b_idx)
        a_val[pos] = b_val * (0.9 + 0.2 * rng.random(n_b))          # a near-copy of b on the shared terms
        others = np.setdiff1d(np.arange(len(a_idx)), pos)
        a_val[others] = 0.02 * rng.random(len(others))
        a_val = (a_val / np.sqrt((a_val * a_val).sum())).astype(dtype)
        s = exact_score(a_idx, a_val, b_idx, b_val, dtype)
        if not 0.45 < float(s) < 1.0:
            continue
        thr = np.nextafter(s, dtype(0))                   # score > thr by one ulp: a match
        bq = q8_quantise(b_val, norm_up)
        assert (bq.astype(np.float64) * float(norm_up) / 255.0 >= b_val.astype(np.float64)).all()      # rounded UP
        assert q8_bound(a_idx, a_val, b_idx, bq) >= q8_bar(thr, norm_up), (trial, float(s))
        checked += 1
    assert checked > 300


def test_second_filter_record_layout_restated():
    """What K3 writes for a row (sg_postings.hip: q8_write_unit / sg_q8_units) -- units in use, the flag of rows without a
    copy -- as the kernel reads it (q8_passes: units = (n + 7) >> 2, whole = bit 31 clear)."""
    for nnz, units, whole in ((0, 1, True), (1, 2, True), (4, 2, True), (5, 3, True), (28, 8, True), (29, 9, True),
                              (60, 16, True), (61, 1, False), (500, 1, False)):
        k3_units = 1 if nnz > Q8_MAX_ENTRIES else (nnz + 7) >> 2
        word2 = nnz | (0 if nnz <= Q8_MAX_ENTRIES else 0x80000000)
        kernel_whole = (word2 >> 31) == 0
        kernel_units = ((word2 & 0x7fffffff) + 7) >> 2 if kernel_whole else 1
        assert (k3_units, kernel_whole, kernel_units) == (units, whole, units)
        assert units * 16 <= 256 and (not whole or 4 * (units - 1) >= nnz)
