// K4p -- the top-n multiply with exact prefix-filter pruning (cosine data: non-negative values, row
// norms <= 1, i.e. what the TF-IDF vectoriser produces; anything else runs the exact kernel K4).
//
// Same contract as K4 (sparse_dot_topn.sp_matmul_topn as called from string_grouper/string_grouper.py
// :725-732 of the reference) and bit-identical results; it only avoids work that provably cannot
// produce a score above the threshold.  The survey lists this as the remedy for the skew of name data
// (hard part 3): a handful of n-grams ('inc', 'llc', ' co') are present in a fifth
// of all rows and generate most of the intermediate products, yet carry almost no weight.
//
// For left row i (non-zeros a_k) the terms are split into a suffix S -- the most frequent terms, all of
// them "frequent" (list length >= freq_min), as many as fit under ||a_S|| * max_j ||b_j|| <= beta =
// threshold - delta -- and a prefix P (the rest: the rare terms).  With f_j = ||b_j restricted to the
// frequent terms||, Cauchy-Schwarz gives
//        score(i, j) = sum_{k in P} a_k b_jk + sum_{k in S} a_k b_jk <= p_ij + ||a_S|| f_j <= p_ij + beta
// so (1) a right row j that shares no term of P with row i cannot reach the threshold -- only the
// posting lists of P are streamed (4-25x fewer entries than all lists), and (2) of the rows that do,
// only those with p_ij > threshold - ||a_S|| f_j need their score computed.  Those "survivors" are
// scored EXACTLY -- row j of B is walked in ascending k, every term looked up in row i, product and sum
// rounded separately: the reference's arithmetic and summation order -- then the strict threshold,
// the canonical order (score desc, column asc) and the top-n cut are applied as in K4.
//
// p_ij is only a filter: any upper bound will do, in any summation order.  K3 therefore writes, next
// to the postings proper, 4-byte "filter postings" {column in tile: L bits, bq: 24 - L, fq: 8} with the
// value b and the norm f_j quantised UPWARDS, and the kernel accumulates upper bounds of a * b in
// 16-bit fixed point (scale 2^15) with LDS integer atomics.  That removes K4's structural cost -- one
// wave instruction per posting segment because two segments may hit the same column -- here every
// lane of the wave carries a posting of whatever term: lanes are dealt to the terms of P in
// proportion to their list lengths, lane (term g, u of G_g) walks postings lo_g + u, lo_g + u + G_g, ...
// of the current column tile.  ds_add_rtn_u32 returns the previous value, so the lane whose add takes
// an accumulator across its column's survivor threshold knows it (values only grow) and appends the
// column to the wave's survivor list -- no sweep of the tile, it is just cleared.  Measured on MI355X:
// 7.6-13 cycles per 64-lane ds_add(_rtn)_u32 per CU against 16.5 for a plain read-add-write and 212
// for ds_add_f32 (profiles/r01_lds_atomic_microbench.log).
//
// One 64-lane wave per left row, single-wave workgroups, persistent waves fed by a global row
// counter, as K4.  LDS per wave: the tile's 4096 u16 accumulators (two per word), row i as a term ->
// value hash (for the exact scoring) and the survivor buffer: 10 KiB -> 16 waves per CU.
// Rows the kernel does not handle (more than 64 non-zeros) are appended to a list and processed by K4.
#define SG_WATCH_NAME sg_debug_watch_pruned
#include "sg_k4_device.h"

#define SG_SURV_CAP 128   // survivors buffered per wave (verified 64 at a time as soon as 64 are there)

template <typename T>
__device__ __forceinline__ T wave_shfl(T v, int src) {
    return __shfl(v, src, 64);
}

// Row i of A is staged in LDS as a 128-slot open-addressing hash (term -> value) so that a lane can walk
// row j of B with INDEPENDENT loads -- eight entries in flight per round, not one dependent load per merge
// step -- and look every term up in one or two LDS reads.
#define SG_HASH_SLOTS 128
__device__ __forceinline__ uint32_t term_hash(int k) { return ((uint32_t)k * 2654435761u) >> 25; }

// Packed rows of B (sg_postings.hip, fwd_pack): eight entries per round, as 16-byte loads.
template <typename T>
struct FwdRound;
template <>
struct FwdRound<float> {   // 4 loads x 2 entries; q is even
    int k[8];
    float v[8];
    __device__ __forceinline__ void load(const void *fwd, uint32_t q, uint32_t last) {
        const uint4 *f = reinterpret_cast<const uint4 *>(fwd);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint4 w = f[min((q >> 1) + e, last >> 1)];
            k[2 * e] = (int)w.x;
            v[2 * e] = __uint_as_float(w.y);
            k[2 * e + 1] = (int)w.z;
            v[2 * e + 1] = __uint_as_float(w.w);
        }
    }
};
template <>
struct FwdRound<double> {   // 8 loads x 1 entry
    int k[8];
    double v[8];
    __device__ __forceinline__ void load(const void *fwd, uint32_t q, uint32_t last) {
        const uint4 *f = reinterpret_cast<const uint4 *>(fwd);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const uint4 w = f[min(q + e, last)];
            k[e] = (int)w.x;
            v[e] = __longlong_as_double((long long)(((unsigned long long)w.w << 32) | w.z));
        }
    }
};

// Exact score of (row i of A, row j of B) for the lanes with j >= 0: the products of the shared terms are
// added in ascending k (B's rows are sorted), product and sum rounded separately -- the reference's
// arithmetic.  A term row i does not have contributes a * b with a = 0: sum + 0 == sum exactly (all
// values are non-negative), so absent terms and the padding of a round need no branch.  The hits
// (score > threshold) go into the register top-n list.
template <typename T>
__device__ __forceinline__ void verify_chunk(int j, const int *hk, const T *ha, const uint32_t *__restrict__ fwd_ptr,
                                             const void *__restrict__ fwd, T thr, TopList<T> &top, int lane) {
    T sum = (T)0;
    if (j >= 0) {
        const uint32_t pb = fwd_ptr[j];
        const uint32_t pe = fwd_ptr[j + 1];
        SG_WD_DECL(wd_v);
        for (uint32_t q = pb & ~1u; q < pe; q += 8) {
            SG_WD(wd_v, 1 << 20, 21)
            FwdRound<T> r;
            r.load(fwd, q, pe - 1u);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                T a = (T)0;
                if (q + e >= pb && q + e < pe) {
                    uint32_t h = term_hash(r.k[e]);
                    int key = hk[h];
                    SG_WD_DECL(wd_p);
                    while (key != r.k[e] && key != -1) {
                        SG_WD(wd_p, SG_HASH_SLOTS + 2, 23)
                        h = (h + 1) & (SG_HASH_SLOTS - 1);
                        key = hk[h];
                    }
                    if (key == r.k[e]) a = ha[h];
                }
                sum = add_rn<T>(sum, mul_rn<T>(a, r.v[e]));
            }
        }
    }
    uint64_t hm = __ballot(j >= 0 && sum > thr);
    SG_WD_DECL(wd_h);
    while (hm) {
        SG_WD(wd_h, 70, 22)
        const int src = __builtin_ctzll(hm);
        hm &= hm - 1;
        top.insert(wave_read<T>(sum, src), wave_read<int>(j, src), lane);
    }
}

// Four filter postings per lane of one column tile: lane (term g, u of G) holds entries idx, idx + G,
// idx + 2G, idx + 3G of its term's segment.  The loads are unconditional (lanes without an entry re-read
// posting 0) so that a batch can stay in flight while the previous one is applied.
struct FiltBatch {
    uint32_t r[4];
    bool ok[4];
    __device__ __forceinline__ void issue(const uint32_t *__restrict__ filt, uint32_t idx, uint32_t hi, uint32_t g) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            ok[e] = g != 0 && idx + e * g < hi;
            const uint32_t off = (ok[e] ? idx + e * g : 0u) << 2;   // kernel-constant base + 32-bit byte offset
            r[e] = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(filt) + off);
        }
    }
};

template <typename T, int TILE_LOG2>
__global__ void __launch_bounds__(64)
spgemm_topn_pruned_kernel(const int64_t *__restrict__ a_indptr, const int32_t *__restrict__ a_indices,
                          const T *__restrict__ a_data, uint32_t n_left, const uint32_t *__restrict__ seg,
                          const uint32_t *__restrict__ filt, int32_t n_tiles, const uint32_t *__restrict__ fwd_ptr,
                          const void *__restrict__ fwd, int32_t keep, int32_t out_stride, T thr,
                          float s_budget /* (beta / max ||b_j||)^2, rounded down */, float norm_b /* max ||b_j||, rounded up */,
                          uint32_t freq_min /* list length from which a term may join the suffix */,
                          int32_t *__restrict__ out_cols, T *__restrict__ out_vals, int32_t *__restrict__ out_cnt,
                          uint32_t *row_counter, uint32_t *flagged_count, uint32_t *flagged_rows,
                          unsigned long long *stats /* [0] rows [1] postings streamed [2] survivors */) {
    constexpr int TILE = 1 << TILE_LOG2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *tab = reinterpret_cast<uint32_t *>(smem);                       // TILE u16 accumulators
    int *hk = reinterpret_cast<int *>(smem + TILE * 2);                       // row i: hash of its terms
    T *ha = reinterpret_cast<T *>(smem + TILE * 2 + 512);                     //        and their values
    int *surv = reinterpret_cast<int *>(smem + TILE * 2 + 512 + 1024);        // SG_SURV_CAP columns
    uint4 *tab_v = reinterpret_cast<uint4 *>(smem);
    const int lane = threadIdx.x;
    for (int x = lane; x < TILE * 2 / 16; x += 64) tab_v[x] = make_uint4(0, 0, 0, 0);
    const uint64_t lanes_below = (1ull << lane) - 1ull;
    unsigned long long st_rows = 0, st_post = 0, st_surv = 0;

    // rows are handed out four at a time: one global atomic per row capped the kernel at ~88 rows/us
    SG_WD_DECL(wd_rows);
    for (uint32_t row0 = next_row(row_counter, lane) * 4u; row0 < n_left; row0 = next_row(row_counter, lane) * 4u)
    for (uint32_t row = row0; row < min(row0 + 4u, n_left); ++row) {
        SG_WD(wd_rows, n_left + 2, 11)
        const int64_t rlo = a_indptr[row];
        const int nnz = __builtin_amdgcn_readfirstlane((int)(a_indptr[row + 1] - rlo));
        if (nnz > 64) {   // more non-zeros than lanes: exact kernel
            if (lane == 0) flagged_rows[atomicAdd(flagged_count, 1u)] = row;
            continue;
        }
        if (nnz == 0) continue;   // out_cnt is zero-initialised
        ++st_rows;
        int k = 0;
        T a = (T)0;
        uint32_t df = 0;
        if (lane < nnz) {
            k = a_indices[rlo + lane];
            a = a_data[rlo + lane];
            const uint32_t *sp = seg + (int64_t)k * n_tiles;
            df = sp[n_tiles] - sp[0];
        }
        // ---- suffix S: the most frequent terms while the bound on ||a_S|| holds.  cum = sum of squares of
        // the terms ordered before this lane's (list length descending, lane ascending), inclusive.
        const float w = lane < nnz ? (float)a * (float)a * 1.00001f : 0.f;   // rounded up: covers the float sums below
        float cum = 0.f;
        for (int q = 0; q < nnz; ++q) {
            const uint32_t dq = wave_read<uint32_t>(df, q);
            const float wq = wave_read<float>(w, q);
            const bool before = dq > df || (dq == df && q <= lane);
            cum += before ? wq : 0.f;
        }
        const bool in_s = lane < nnz && cum <= s_budget && df >= freq_min;   // a prefix of the (df desc) order
        const bool in_p = lane < nnz && !in_s;
        const uint64_t pm = __ballot(in_p);
        if (pm == 0) continue;   // ||a|| * max ||b|| <= beta < threshold: no match possible
        float bs2 = in_s ? cum : 0.f;
        float dsum = in_p ? (float)df : 0.f;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            bs2 = fmaxf(bs2, __shfl_xor(bs2, d, 64));
            dsum += __shfl_xor(dsum, d, 64);
        }
        bs2 = wave_read<float>(bs2, 0);     // explicitly wave-uniform: the branches below must not diverge
        dsum = wave_read<float>(dsum, 0);
        if (!(dsum > 0.f)) continue;   // every list of P is empty
        st_post += (unsigned long long)dsum;   // statistic (float sum of the list lengths)
        const int np = __popcll(pm);
        // ---- survivor test in fixed point (scale 2^15).  q_ij accumulates UPPER bounds of the products, so
        //      p_ij * 2^15 <= q_ij; the exact kernel's float score obeys  score~ <= score + 1e-5  and
        //      score <= p_ij + ||a_S|| f_j  with  f_j <= fq_j / 255 * norm_b.  Column j survives when
        //      q_ij >= tq_j = (thr - 1e-5) * 2^15 - c1 * fq_j - 2   (the 2 covers the float evaluation).
        const float b_s = sqrtf(bs2) * 1.000002f;
        const float t0 = ((float)thr - 1e-5f) * 32768.0f - 2.0f;
        const float c1 = b_s * norm_b * (32768.0f / 255.0f) * 1.000002f;
        if (!(t0 - c1 * 255.0f >= 1.0f)) {   // delta too small for the fixed-point resolution: exact kernel
            if (lane == 0) flagged_rows[atomicAdd(flagged_count, 1u)] = row;
            continue;
        }

        // ---- deal the 64 lanes to the terms of P in proportion to their list lengths
        uint32_t G = in_p ? 1u + (uint32_t)((float)(64 - np) * 0.999f * ((float)df / dsum)) : 0u;
        uint32_t start = G;   // inclusive scan, made exclusive below
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_up(start, d, 64);
            if (lane >= d) start += o;
        }
        start -= G;
        int src = 0;
        uint32_t u = 0, g = 0;
        {
            uint64_t m = pm;
            SG_WD_DECL(wd_a);
            while (m) {
                SG_WD(wd_a, 70, 12)
                const int f = __builtin_ctzll(m);
                m &= m - 1;
                const uint32_t sf = wave_read<uint32_t>(start, f), gf = wave_read<uint32_t>(G, f);
                const uint32_t d = (uint32_t)lane - sf;
                if (d < gf) {
                    src = f;
                    u = d;
                    g = gf;
                }
            }
        }
        const int my_k = wave_shfl<int>(k, src);
        // upper bound of a * b * 2^15 from the bq of a filter posting: (uint32)(c_a * bq) + 1
        const float c_a = (float)wave_shfl<T>(a, src) * norm_b * (32768.0f / (float)((1 << (24 - TILE_LOG2)) - 1)) * 1.000002f;
        const uint32_t seg_off = (uint32_t)my_k * (uint32_t)n_tiles;   // < 2^30 bins (checked when the postings are built)
        auto seg_at = [&](uint32_t i) {
            return *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(seg) + ((seg_off + i) << 2));
        };

        // ---- stage row i for the exact scoring: term -> value hash (filled by compare-and-swap, one wave)
        hk[lane] = -1;
        hk[lane + 64] = -1;
        __builtin_amdgcn_wave_barrier();
        if (lane < nnz) {
            uint32_t h = term_hash(k);
            SG_WD_DECL(wd_i);
            while (atomicCAS(&hk[h], -1, k) != -1) {
                SG_WD(wd_i, SG_HASH_SLOTS + 2, 16)
                h = (h + 1) & (SG_HASH_SLOTS - 1);
            }
            ha[h] = a;
        }
        __builtin_amdgcn_wave_barrier();

        TopList<T> top;
        top.clear();
        uint32_t n_surv = 0;
        // applies one batch to the tile's accumulators; a lane whose add takes an accumulator across tq
        // appends the column to the survivor buffer; full waves of survivors are scored at once
        auto apply = [&](const FiltBatch &bt, int t) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (e > 0 && __ballot(bt.ok[e]) == 0) break;   // a lane's entries fill its slots in order
                bool cross = false;
                uint32_t c = 0;
                if (bt.ok[e]) {
                    c = bt.r[e] & (uint32_t)(TILE - 1);
                    const uint32_t x = (uint32_t)(c_a * (float)((bt.r[e] >> TILE_LOG2) & ((1u << (24 - TILE_LOG2)) - 1u))) + 1u;
                    const uint32_t tq = (uint32_t)(t0 - c1 * (float)(bt.r[e] >> 24));
                    const uint32_t sh = (c & 1u) << 4;
                    const uint32_t old = __hip_atomic_fetch_add(&tab[c >> 1], x << sh, __ATOMIC_RELAXED,
                                                                __HIP_MEMORY_SCOPE_WORKGROUP);
                    const uint32_t oh = (old >> sh) & 0xffffu;
                    cross = oh < tq && oh + x >= tq;
                }
                const uint64_t cm = __ballot(cross);
                if (cm) {
                    if (cross) surv[n_surv + __popcll(cm & lanes_below)] = (t << TILE_LOG2) + (int)c;
                    n_surv += __popcll(cm);
                    if (n_surv >= 64) {   // a full wave of survivors: score them now (the buffer holds 128)
                        verify_chunk<T>(surv[lane], hk, ha, fwd_ptr, fwd, thr, top, lane);
                        st_surv += 64;
                        const uint32_t rem = n_surv - 64;   // < 64: move to the front
                        const int keepv = (uint32_t)lane < rem ? surv[64 + lane] : 0;
                        __builtin_amdgcn_wave_barrier();
                        if ((uint32_t)lane < rem) surv[lane] = keepv;
                        n_surv = rem;
                    }
                }
            }
        };

        // Software pipeline over the column tiles: while tile t is applied, the first batches of tiles t + 1
        // and t + 2 are in flight and the segment bound needed for tile t + 3 is being fetched (b0..b4 =
        // seg[t .. t + 4] of my term).  The multiply is bound by load latency (~1-2 us to the Infinity
        // Cache at 14 waves per CU): with one tile of prefetch instead of two the kernel takes 36.1 ms instead of 33.5.
        const uint32_t last = (uint32_t)n_tiles;
        uint32_t b0 = 0, b1 = 0, b2 = 0, b3 = 0, b4 = 0;
        if (g) {
            b0 = seg_at(0);
            b1 = seg_at(min(1u, last));
            b2 = seg_at(min(2u, last));
            b3 = seg_at(min(3u, last));
        }
        FiltBatch cur, nx1;
        cur.issue(filt, b0 + u, b1, g);
        nx1.issue(filt, b1 + u, b2, g);
        SG_WD_DECL(wd_t);
        for (int t = 0; t < n_tiles; ++t) {
            SG_WD(wd_t, n_tiles + 2, 13)
            if (g) b4 = seg_at(min((uint32_t)t + 4u, last));
            FiltBatch nx2;
            nx2.issue(filt, b2 + u, b3, g);   // past the last tile the bounds coincide: nothing to load
            if (__ballot(cur.ok[0]) != 0) {
                apply(cur, t);
                uint32_t idx = b0 + u + 4 * g;
                if (__ballot(g != 0 && idx < b1) == 0) {
                    // re-zero only what was touched (a few dozen of the tile's accumulators): sweeping the
                    // tile for every row costs rows * columns * 2 B of LDS writes, 11 ms at 663 k
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (cur.ok[e]) tab[(cur.r[e] & (uint32_t)(TILE - 1)) >> 1] = 0u;
                } else {
                    SG_WD_DECL(wd_b);
                    do {   // segments longer than the dealt lanes cover in one batch
                        SG_WD(wd_b, 1 << 24, 14)
                        FiltBatch more;
                        more.issue(filt, idx, b1, g);
                        apply(more, t);
                        idx += 4 * g;
                    } while (__ballot(g != 0 && idx < b1) != 0);
                    for (int x = lane; x < TILE * 2 / 16; x += 64) tab_v[x] = make_uint4(0, 0, 0, 0);
                }
            }
            b0 = b1;
            b1 = b2;
            b2 = b3;
            b3 = b4;
            cur = nx1;
            nx1 = nx2;
        }
        if (n_surv > 0) {   // fewer than 64 left
            verify_chunk<T>((uint32_t)lane < n_surv ? surv[lane] : -1, hk, ha, fwd_ptr, fwd, thr, top, lane);
            st_surv += n_surv;
        }
        int cnt = __popcll(__ballot(top.c != INT32_MAX));
        if (cnt > keep) cnt = keep;
        const size_t obase = (size_t)row * (size_t)out_stride;
        if (lane < cnt) {
            out_vals[obase + lane] = top.s;
            out_cols[obase + lane] = top.c;
        }
        if (lane == 0) out_cnt[row] = cnt;
    }
    if (lane == 0) {
        if (st_rows) atomicAdd(stats + 0, st_rows);
        if (st_post) atomicAdd(stats + 1, st_post);
        if (st_surv) atomicAdd(stats + 2, st_surv);
    }
}

// ------------------------------------------------------------------------------------------------
// Properties a matrix must have for the pruned kernel: values >= 0 (no NaN), column indices strictly
// ascending within every row, row norms <= 1 (+ rounding).  One thread per row.
template <typename T>
__global__ void __launch_bounds__(256) csr_props_kernel(const int64_t *__restrict__ indptr,
                                                        const int32_t *__restrict__ indices,
                                                        const T *__restrict__ data, int64_t n_rows,
                                                        uint32_t *out /* [0] violations [1] max ||row||^2 as float bits */) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t bad = 0;
    float n2 = 0.f;
    if (i < n_rows) {
        double s = 0.0;
        int prev = -1;
        for (int64_t p = indptr[i]; p < indptr[i + 1]; ++p) {
            const T v = data[p];
            const int k = indices[p];
            if (!(v >= (T)0)) bad = 1;
            if (k <= prev) bad = 1;
            prev = k;
            s += (double)v * (double)v;
        }
        n2 = __double2float_ru(s);
    }
    uint32_t nb = __float_as_uint(n2);   // non-negative floats order like unsigned integers
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        nb = max(nb, (uint32_t)__shfl_xor((int)nb, d, 64));
        bad |= (uint32_t)__shfl_xor((int)bad, d, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        if (bad) atomicOr(out, 1u);
        atomicMax(out + 1, nb);
    }
}

int sg_csr_props(sg_ctx *ctx, const sg_csr *m, bool *cosine_like, float *max_norm2) {
    if (m->props_state == 0) {
        uint32_t *d = nullptr;
        SG_TRY(sg_alloc(ctx, (size_t)2, &d));
        uint32_t h[2] = {0, 0};
        hipError_t e = hipMemsetAsync(d, 0, 8, ctx->stream);
        if (e == hipSuccess && m->n_rows > 0) {
            const unsigned grid = (unsigned)((m->n_rows + 255) / 256);
            if (m->dtype == SG_F64)
                hipLaunchKernelGGL(csr_props_kernel<double>, dim3(grid), dim3(256), 0, ctx->stream, m->d_indptr,
                                   m->d_indices, (const double *)m->d_data, m->n_rows, d);
            else
                hipLaunchKernelGGL(csr_props_kernel<float>, dim3(grid), dim3(256), 0, ctx->stream, m->d_indptr,
                                   m->d_indices, (const float *)m->d_data, m->n_rows, d);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(h, d, 8, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        ctx->release(d);
        if (e != hipSuccess) {
            sg_set_error("sg_csr_props: %s", hipGetErrorString(e));
            return SG_ERR_HIP;
        }
        float n2;
        memcpy(&n2, &h[1], 4);
        m->props_max_norm2 = n2;
        m->props_state = (h[0] == 0 && n2 <= 1.0001f) ? 1 : 2;
    }
    *cosine_like = m->props_state == 1;
    *max_norm2 = m->props_max_norm2;
    return SG_OK;
}

// ------------------------------------------------------------------------------------------------
template <typename T, int TILE_LOG2>
static int launch_pruned(sg_ctx *ctx, const sg_csr *A, const sg_postings *Bt, int32_t keep, sg_topn *r, T thr,
                         float s_budget, uint32_t *row_counter, uint32_t *flagged_count, uint32_t *flagged_rows,
                         unsigned long long *stats) {
    const size_t lds = ((size_t)2 << TILE_LOG2) + 512 + 1024 + (size_t)SG_SURV_CAP * 4;
    int waves_per_cu = (int)(ctx->lds_per_cu / lds);
    if (waves_per_cu > 32) waves_per_cu = 32;
    if (waves_per_cu < 1) waves_per_cu = 1;
    if (const char *v = getenv("SG_PRUNE_WAVES_PER_CU"))
        if (atoi(v) > 0) waves_per_cu = atoi(v);
    unsigned grid = (unsigned)ctx->num_cu * (unsigned)waves_per_cu;
    if ((int64_t)grid > A->n_rows) grid = (unsigned)(A->n_rows > 0 ? A->n_rows : 1);
    hipLaunchKernelGGL((spgemm_topn_pruned_kernel<T, TILE_LOG2>), dim3(grid), dim3(64), lds, ctx->stream, A->d_indptr,
                       A->d_indices, (const T *)A->d_data, (uint32_t)A->n_rows, (const uint32_t *)Bt->d_seg,
                       (const uint32_t *)Bt->d_filt, Bt->n_tiles, (const uint32_t *)Bt->d_fwd_ptr,
                       (const void *)Bt->d_fwd, keep, r->stride, thr, s_budget, Bt->norm_up, Bt->freq_min, r->d_cols,
                       (T *)r->d_vals,
                       r->d_counts, row_counter, flagged_count, flagged_rows, stats);
    SG_HIP_TRY(hipGetLastError());
    return SG_OK;
}

template <typename T>
static int dispatch_pruned(sg_ctx *ctx, const sg_csr *A, const sg_postings *Bt, int32_t keep, sg_topn *r, T thr,
                           float s_budget, uint32_t *row_counter, uint32_t *flagged_count, uint32_t *flagged_rows,
                           unsigned long long *stats) {
    switch (Bt->tile_log2) {
        case 11: return launch_pruned<T, 11>(ctx, A, Bt, keep, r, thr, s_budget, row_counter, flagged_count, flagged_rows, stats);
        case 12: return launch_pruned<T, 12>(ctx, A, Bt, keep, r, thr, s_budget, row_counter, flagged_count, flagged_rows, stats);
        case 13: return launch_pruned<T, 13>(ctx, A, Bt, keep, r, thr, s_budget, row_counter, flagged_count, flagged_rows, stats);
        default:
            sg_set_error("postings tile of 2^%d columns is not supported by the pruned multiply (2^11..2^13)", Bt->tile_log2);
            return SG_ERR_UNSUPPORTED;
    }
}

bool sg_pruned_supports_tile(int32_t tile_log2) { return tile_log2 >= 11 && tile_log2 <= 13; }

int sg_spgemm_pruned_launch(sg_ctx *ctx, const sg_csr *A, const sg_postings *Bt, int32_t keep, sg_topn *r,
                            double threshold, double delta, uint32_t *row_counter, uint32_t *flagged_count,
                            uint32_t *flagged_rows, unsigned long long *stats) {
    // beta = threshold - delta bounds ||a_S|| * max ||b_j||; the kernel compares sums of squares of a
    const double nb = (double)Bt->norm_up;
    const double beta = threshold - delta;
    const double budget = (beta / nb) * (beta / nb) * (1.0 - 1e-6);
    const float s_budget = __builtin_nextafterf((float)budget, 0.f);
    if (A->dtype == SG_F64)
        return dispatch_pruned<double>(ctx, A, Bt, keep, r, (double)threshold, s_budget, row_counter, flagged_count,
                                       flagged_rows, stats);
    return dispatch_pruned<float>(ctx, A, Bt, keep, r, (float)threshold, s_budget, row_counter, flagged_count,
                                  flagged_rows, stats);
}
