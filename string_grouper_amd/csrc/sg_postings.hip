// K3 -- inverted index of the right-hand matrix, bucketed by column tile.
//
// Replaces what the reference does at string_grouper/string_grouper.py:727 / :738
// (duplicate_matrix.transpose(), then sparse_dot_topn's internal CSC->CSR conversion of B^T).
//
// Layout in HBM.  For every term k (a column of the TF-IDF matrices) and every tile t of
// tile_cols = 2^tile_log2 consecutive right-hand rows, the postings
//     { (j, B[j,k]) : j in tile t }
// are stored contiguously; segment (k, t) is [seg[k*n_tiles+t], seg[k*n_tiles+t+1]).  The multiply
// (K4) walks the tiles of one left row with a private LDS accumulator of tile_cols values, so a tile
// only ever streams its own slice of each posting list.  Inside a segment the order of the j is
// irrelevant to the result (each (i,j) accumulator receives exactly one product per k), so the
// scatter below uses atomic cursors and needs no sort: a two-pass counting sort keyed by (k, tile).
//
// Bound: HBM.  Algorithmic bytes = 2 * nnz * (4 + s) + 4 * (V * n_tiles + n) (read B twice,
// write postings once, histogram + scan of the segment table).
#include <stdlib.h>

#include "sg_internal.h"
#include "sg_scan.h"

// Thread -> right-hand row.  Consecutive rows share a column tile, hence the bins of the frequent terms:
// with thread i on row i a whole wave hammers the same counter.  Consecutive threads are therefore dealt
// to consecutive TILES (thread g -> row (g mod n_tiles) * tile + g div n_tiles); every thread still walks
// its own row, so nothing is lost in coalescing.
__device__ __forceinline__ int64_t row_of_thread(int64_t g, int32_t tile_log2, int32_t n_tiles) {
    return ((g % n_tiles) << tile_log2) | (g / n_tiles);
}

template <typename T>
__global__ void __launch_bounds__(256) postings_count(const int64_t *__restrict__ indptr,
                                                      const int32_t *__restrict__ indices, int64_t n_rows,
                                                      int32_t tile_log2, int32_t n_tiles, uint32_t *seg_counts) {
    // one wave per 64 rows would leave lanes idle on short rows; nnz is only ~19/row, so a thread per
    // row with a short serial loop keeps the code simple
    const int64_t j = row_of_thread((int64_t)blockIdx.x * blockDim.x + threadIdx.x, tile_log2, n_tiles);
    if (j >= n_rows) return;
    const int64_t lo = indptr[j], hi = indptr[j + 1];
    const uint32_t t = (uint32_t)(j >> tile_log2);
    for (int64_t p = lo; p < hi; ++p) {
        const int64_t bin = (int64_t)indices[p] * n_tiles + t;
        atomicAdd(&seg_counts[bin], 1u);
    }
}

// posting entry layout (read by Post<T>::load in sg_spgemm_topn.hip):
//   f32: packed {uint32 slot, float value}, 8 bytes, in the vals array (rows array unused)
//   f64: slots[] (uint32, in the rows array) + vals[] (double)
// slot = (j mod tile_cols) * sizeof(T), the byte offset of column j's accumulator in the LDS tile
template <typename T>
__device__ __forceinline__ void store_posting(int32_t *rows, T *vals, uint32_t pos, int32_t j, T v);
template <>
__device__ __forceinline__ void store_posting<float>(int32_t *, float *vals, uint32_t pos, int32_t j, float v) {
    reinterpret_cast<uint2 *>(vals)[pos] = make_uint2((uint32_t)j, __float_as_uint(v));
}
template <>
__device__ __forceinline__ void store_posting<double>(int32_t *rows, double *vals, uint32_t pos, int32_t j, double v) {
    rows[pos] = j;
    vals[pos] = v;
}

// norm of a row's frequent part (terms whose list holds >= freq_min entries) relative to norm_up, rounded up
// (emit_posting quantises it upwards once more, to the bits the form of the posting has for it)
__device__ __forceinline__ float frequent_norm_ratio_of(double f2, float inv_norm_up) {
    return __double2float_ru(sqrt(f2) * (1.0 + 1e-12)) * inv_norm_up;
}
template <typename T>
__device__ __forceinline__ float frequent_norm_ratio(const int32_t *__restrict__ indices, const T *__restrict__ data,
                                                     int64_t lo, int64_t hi, const uint32_t *__restrict__ seg, int32_t n_tiles,
                                                     uint32_t freq_min, float inv_norm_up) {
    double f2 = 0.0;
    for (int64_t p = lo; p < hi; ++p) {
        const int64_t k = indices[p];
        if (seg[(k + 1) * n_tiles] - seg[k * n_tiles] >= freq_min) f2 += (double)data[p] * (double)data[p];
    }
    return frequent_norm_ratio_of(f2, inv_norm_up);
}

// Entry i of the 512 "null postings" behind the filter postings (stream form: a lane that is through with its segment
// reads entries 4 * lane .. 4 * lane + 3): bq = fq = 0 and ALL FOUR entries of a lane name accumulator word `lane` -- one
// LDS instruction of the wave then touches 64 different words, and the low 24 bits of the entry, which the multiply
// takes for the value, stay below 256: (CA * 252) >> 32 == 0 for every CA < 2^24, the entry adds nothing.
__device__ __forceinline__ uint32_t sg_null_posting(uint32_t i) { return ((i >> 2) << 2) | 0x80000000u; }

// one posting (+ its filter posting) of column `col` of the tile at position `pos`.
// Filter posting (read by K4p, sg_spgemm_pruned.hip), 32 bits, AB = tile_log2 + 1.  `fr` = the norm of the row's frequent
// part relative to norm_up, rounded up (frequent_norm_ratio).
// Tile-by-tile form (fold_log2 == 0):
//   [0]        h     which 16-bit half of the accumulator word the column owns (col & 1)
//   [1]        0
//   [2, AB)    word  (col >> 1): bits [0, AB) masked with ~3 ARE the byte address of the accumulator word in LDS
//   [AB, 24)   bq    value quantised upwards relative to norm_up
//   [24, 32)   fq    fr quantised upwards to 8 bits
// Stream form (fold_log2 == 3, tile_log2 == 12; sg_spgemm_pruned.hip, "stream form"): 2^fold_log2 consecutive tiles share
// ONE accumulator tile, the posting says which of them its column lies in, and the fields are cut for the multiply's
// instructions (round 4: 11 -> 8 VALU per posting):
//   [0]        0
//   [1]        h     (<< 3 it is the shift of the half: 0 or 16; bits [1, 16) >> 1 ARE the column inside the super-tile)
//   [2, 13)    word
//   [13, 16)   fold  tile index mod 8
//   [16, 24)   bq    chosen so that the low 24 bits AS ONE NUMBER are >= v / norm_up * 255 * 2^16: the multiply feeds the
//                    posting to v_mul_hi_u32_u24 as it is -- the bits below bq then count as part of the value, and K3,
//                    which knows them, rounds bq down by what they are worth (the bound is as tight as rounding bq up
//                    by itself was, and the multiply saves the mask)
//   [24, 32)   fq    chosen the same way: bits [16, 32) as one number F >= fr * 255 * 2^8 (bq counts as its low byte), stored
//                    with its top bit flipped (F - 32768 as int16) for v_mad_i32_i16, which then gives the column's whole
//                    survivor threshold in ONE instruction (the 8-bit form: byte select + multiply, subtract, shift)
#define SG_FILT_F16_MAX 65280u   // 255 * 256
// the 32-bit filter posting of value v in column `col` of tile `tile` (the layouts above)
__device__ __forceinline__ uint32_t filter_posting(uint32_t col, float v, float fr, int32_t tile_log2, float inv_norm_up, uint32_t tile,
                                                   int32_t fold_log2) {
    const int32_t ab = tile_log2 + 1, fb = ab + fold_log2;
    if (fold_log2 > 0) {   // stream form (fb == 16: sg_postings_build only folds tiles of 4096 columns by 8)
        const uint32_t low = ((col >> 1) << 2) | ((col & 1u) << 1) | ((tile & ((1u << fold_log2) - 1u)) << ab);
        uint32_t b24 = (uint32_t)ceilf(v * inv_norm_up * (255.0f * 65536.0f) * 1.000002f);
        // v <= norm_up: what lies above 255 * 2^16 is the safety factor's doing.  (Without the cut a row of ONE term --
        // v = 1 -- in one of a tile's first columns, where `low` is smaller than that excess, got bq = 256: the field
        // wrapped to 0 and carried into fq, and the row did not find itself.)
        if (b24 > (255u << 16)) b24 = 255u << 16;
        const uint32_t bq = b24 > low ? (b24 - low + 65535u) >> 16 : 0u;                   // <= 255
        uint32_t f16 = (uint32_t)ceilf(fr * (float)SG_FILT_F16_MAX * 1.000002f);
        if (f16 > SG_FILT_F16_MAX) f16 = SG_FILT_F16_MAX;
        const uint32_t fq = f16 > bq ? (f16 - bq + 255u) >> 8 : 0u;                        // <= 255
        return low | (bq << 16) | ((fq ^ 0x80u) << 24);
    }
    const uint32_t bq_max = (1u << (24 - fb)) - 1u;   // the bits the address and fq leave
    uint32_t bq = (uint32_t)ceilf(v * inv_norm_up * (float)bq_max * 1.000002f);
    if (bq > bq_max) bq = bq_max;
    uint32_t fq = (uint32_t)ceilf(fr * 255.0f * 1.000002f);
    if (fq > 255u) fq = 255u;
    return ((col >> 1) << 2) | (col & 1u) | (bq << fb) | (fq << 24);
}
template <typename T>
__device__ __forceinline__ void emit_posting(int32_t *out_rows, T *out_vals, uint32_t *out_filt, uint32_t pos, uint32_t col,
                                             T v, float fr, int32_t tile_log2, float inv_norm_up, uint32_t tile,
                                             int32_t fold_log2) {
    // the multiply wants the byte offset of the accumulator inside its LDS tile, not j itself
    if (out_vals) store_posting<T>(out_rows, out_vals, pos, (int32_t)(col * (uint32_t)sizeof(T)), v);   // (null: filter postings only)
    if (out_filt) out_filt[pos] = filter_posting(col, (float)v, fr, tile_log2, inv_norm_up, tile, fold_log2);
}

template <typename T>
__global__ void __launch_bounds__(256) postings_fill(const int64_t *__restrict__ indptr,
                                                     const int32_t *__restrict__ indices,
                                                     const T *__restrict__ data, int64_t n_rows, int32_t tile_log2,
                                                     int32_t n_tiles, const uint32_t *__restrict__ seg,
                                                     uint32_t *cursor, int32_t *out_rows, T *out_vals,
                                                     uint32_t *out_filt /* null: no filter postings */,
                                                     uint32_t freq_min, float inv_norm_up, int32_t fold_log2) {
    const int64_t j = row_of_thread((int64_t)blockIdx.x * blockDim.x + threadIdx.x, tile_log2, n_tiles);
    if (j >= n_rows) return;
    const int64_t lo = indptr[j], hi = indptr[j + 1];
    const uint32_t t = (uint32_t)(j >> tile_log2);
    const uint32_t col = (uint32_t)(j & (((int64_t)1 << tile_log2) - 1));
    const float fq = out_filt ? frequent_norm_ratio<T>(indices, data, lo, hi, seg, n_tiles, freq_min, inv_norm_up) : 0.f;
    for (int64_t p = lo; p < hi; ++p) {
        const int64_t bin = (int64_t)indices[p] * n_tiles + t;
        const uint32_t pos = seg[bin] + atomicAdd(&cursor[bin], 1u);
        emit_posting<T>(out_rows, out_vals, out_filt, pos, col, data[p], fq, tile_log2, inv_norm_up, t, fold_log2);
    }
}

// The same two passes with the counters in LDS (4 bytes per term; used when the vocabulary fits: n_terms * 4 <= 120 KiB):
// a workgroup owns a column tile -- or, with `split` > 1, one of `split` equal parts of it, so that 162 tiles still fill
// 256 CUs; a segment (k, t) is then the parts' sub-segments back to back, which the multiply never notices since the
// order inside a segment is free.  Sixteen lanes walk one row (the loads of a row are contiguous across lanes; a thread
// per row steps 64 rows with one cache line each), LDS atomics take the place of 2 x nnz global ones.
//
// Round 4: the counters live in a table that is WORKGROUP-major while the index is built -- cnt[w * n_terms + k], w =
// tile * split + part -- so that a workgroup writes its histogram and reads its cursors as ONE contiguous run of
// n_terms words.  (Rounds 2-3 kept it term-major, (k * n_tiles + t) * split + part: every workgroup then gathered its
// 18 300 cursors from 18 300 different cache lines, and so did the histogram's write-back.)  The offsets come in two
// steps: postings_colscan_kernel turns every term's column of counts into running sums over w (the term's entries in
// earlier parts) and its total, one small scan over the totals gives the terms' starts, and a cursor is the sum of
// the two.  postings_tables_kernel then writes what the multiply reads -- the term-major segment table and the two
// padded tables of segment ends -- transposed through LDS.
// Both passes walk FOUR rows per sixteen lanes and trip with all their loads issued before the first is used: a
// workgroup's trips are serial (32 of them for a part of 2048 rows) and a trip is a chain of dependent round trips --
// row pointers, entries, LDS, store -- so that the build ran at the latency of its loads, not at any bandwidth
// (profiles/r04_s1_kernel_stats_baseline.csv: 0.31 ms for 100 MB).
#define SG_POST_ROWS 4    // rows per sixteen lanes and trip
__global__ void __launch_bounds__(1024) postings_count_lds(const int64_t *__restrict__ indptr,
                                                           const int32_t *__restrict__ indices, int64_t n_rows,
                                                           int32_t tile_log2, int32_t n_terms, int32_t split,
                                                           uint32_t *__restrict__ cnt /* [wgs][n_terms] */) {
    extern __shared__ uint32_t hist[];
    const int64_t t = blockIdx.x / split;
    const int32_t part = blockIdx.x % split;
    for (int k = threadIdx.x; k < n_terms; k += blockDim.x) hist[k] = 0;
    __syncthreads();
    const int64_t part_rows = ((int64_t)1 << tile_log2) / split;
    const int64_t j0 = (t << tile_log2) + part * part_rows;
    int64_t j1 = j0 + part_rows;
    if (j1 > n_rows) j1 = n_rows;
    const int sub = threadIdx.x & 15;
    const int64_t groups = blockDim.x >> 4;
    for (int64_t jb = j0 + (threadIdx.x >> 4) * SG_POST_ROWS; jb < j1; jb += groups * SG_POST_ROWS) {
        int64_t lo[SG_POST_ROWS], hi[SG_POST_ROWS];
#pragma unroll
        for (int r = 0; r < SG_POST_ROWS; ++r) {
            const bool valid = jb + r < j1;
            lo[r] = valid ? indptr[jb + r] : 0;
            hi[r] = valid ? indptr[jb + r + 1] : 0;
        }
        int32_t k0[SG_POST_ROWS];
#pragma unroll
        for (int r = 0; r < SG_POST_ROWS; ++r) k0[r] = lo[r] + sub < hi[r] ? indices[lo[r] + sub] : -1;
#pragma unroll
        for (int r = 0; r < SG_POST_ROWS; ++r) {
            if (k0[r] >= 0) atomicAdd(&hist[k0[r]], 1u);
            for (int64_t p = lo[r] + sub + 16; p < hi[r]; p += 16) atomicAdd(&hist[indices[p]], 1u);   // rows beyond 16 entries
        }
    }
    __syncthreads();
    uint32_t *mine = cnt + (int64_t)blockIdx.x * n_terms;
    for (int k = threadIdx.x; k < n_terms; k += blockDim.x) mine[k] = hist[k];
}

// Per term: counts of the workgroups -> entries of the term in EARLIER workgroups (in place), the term's total, and
// whether it is frequent.  A workgroup takes 64 terms x all workgroups of the build: 16 threads per term, each a run of
// consecutive w, the runs joined through LDS; loads and stores are contiguous across the 64 terms.
__global__ void __launch_bounds__(1024) postings_colscan_kernel(uint32_t *__restrict__ cnt, int32_t n_wgs, int32_t n_terms,
                                                                uint32_t freq_min, uint32_t *__restrict__ term_len,
                                                                uint8_t *__restrict__ is_frequent) {
    __shared__ uint32_t part_sum[16][64];
    const int kk = threadIdx.x & 63, c = threadIdx.x >> 6;
    const int64_t k = (int64_t)blockIdx.x * 64 + kk;
    const int32_t run = (n_wgs + 15) / 16;
    const int32_t w0 = c * run, w1 = min(w0 + run, n_wgs);
    uint32_t s = 0;
    if (k < n_terms)
        for (int32_t w = w0; w < w1; ++w) s += cnt[(int64_t)w * n_terms + k];
    part_sum[c][kk] = s;
    __syncthreads();
    uint32_t before = 0, total = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const uint32_t v = part_sum[q][kk];
        before += q < c ? v : 0u;
        total += v;
    }
    if (k < n_terms) {
        for (int32_t w = w0; w < w1; ++w) {
            const uint32_t v = cnt[(int64_t)w * n_terms + k];
            cnt[(int64_t)w * n_terms + k] = before;
            before += v;
        }
        if (c == 0) {
            term_len[k] = total;
            if (is_frequent) is_frequent[k] = total >= freq_min ? 1 : 0;
        }
    }
}

// What the multiply reads, from the build's tables: S(k, t) = term_start[k] + cnt[(t * split) * n_terms + k] is the start
// of term k's postings in tile t, S(k, n_tiles) = term_start[k + 1];
//   seg[k * n_tiles + t] = S(k, t)                                   (+ seg[n_terms * n_tiles] = all postings)
//   ends[k * nt_pad + t] = S(k, min(t, n_tiles - 1) + 1) << 2        byte offsets into the filter postings, rows padded to
//                                                                    a multiple of four tiles; row n_terms: zeros (the
//                                                                    segments of a lane without a term)
//   ends8[k * nv_pad + v] = S(k, min((v + 1) << fold_log2, n_tiles)) << 2   the same for super-tiles (stream form)
// A workgroup takes 64 terms and walks the tiles 64 at a time: read with the lanes along the terms (the table is
// workgroup-major), written with the lanes along the tiles (the multiply's tables are term-major), through LDS.
__global__ void __launch_bounds__(1024) postings_tables_kernel(const uint32_t *__restrict__ cnt, const uint32_t *__restrict__ term_start,
                                                               int32_t n_terms, int32_t n_tiles, int32_t split,
                                                               uint32_t *__restrict__ seg, uint32_t *__restrict__ ends, int32_t nt_pad,
                                                               uint32_t *__restrict__ ends8, int32_t nv_pad, int32_t fold_log2,
                                                               SgScoreCtx sc, SgScoreCtx *__restrict__ sc_out /* null: none */,
                                                               uint32_t *__restrict__ null_slack /* null: none */) {
    __shared__ uint32_t tile[64][66];
    if (blockIdx.x == 0) {   // two one-line kernels of the build ride along: the scoring context as a struct in device memory
        if (sc_out && threadIdx.x == 0) {                                   // (score_ctx_kernel)
            sc_out[0] = sc;
            sc.q8 = nullptr;                                                // [1]: the same without the second filter (the
            sc.q8_scale = 0.f;                                              //      multiply chooses per call: sg_q8_applies)
            sc_out[1] = sc;
        }
        if (null_slack && threadIdx.x < 512) null_slack[threadIdx.x] = sg_null_posting(threadIdx.x);   // (null_postings_kernel)
    }
    const int x = threadIdx.x & 63, y = threadIdx.x >> 6;   // y: 0 .. 15
    const int64_t k0 = (int64_t)blockIdx.x * 64;
    auto S = [&](int64_t k, int64_t t) -> uint32_t {        // k <= n_terms, t <= n_tiles
        if (k >= n_terms) return term_start[n_terms];
        return t >= n_tiles ? term_start[k + 1] : term_start[k] + cnt[(t * split) * (int64_t)n_terms + k];
    };
    const int64_t t_lim = nt_pad > n_tiles ? nt_pad : n_tiles;
    for (int64_t tb = 0; tb < t_lim; tb += 64) {
        // columns tb .. tb + 64 (one more than is written: an end is the next tile's start)
        for (int tt = y; tt < 65; tt += 16) {
            const int64_t k = k0 + x;
            tile[x][tt] = k <= n_terms ? S(k, min(tb + tt, (int64_t)n_tiles)) : 0u;
        }
        __syncthreads();
        for (int kq = y; kq < 64; kq += 16) {
            const int64_t k = k0 + kq, t = tb + x;
            if (k < n_terms && t < n_tiles) seg[k * n_tiles + t] = tile[kq][x];
            if (k == n_terms && t == 0) seg[k * n_tiles] = tile[kq][0];                    // the table's last entry: all postings
            if (ends && k <= n_terms && t < nt_pad)
                ends[k * nt_pad + t] = k == n_terms ? 0u : tile[kq][min(t, (int64_t)n_tiles - 1) + 1 - tb] << 2;
        }
        __syncthreads();
    }
    if (ends8) {
        for (int64_t i = threadIdx.x; i < (int64_t)64 * nv_pad; i += blockDim.x) {
            const int64_t k = k0 + (i & 63), v = i >> 6;
            if (k > n_terms) continue;
            ends8[k * nv_pad + v] = k == n_terms ? 0u : S(k, min((v + 1) << fold_log2, (int64_t)n_tiles)) << 2;
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(1024) postings_fill_lds(const int64_t *__restrict__ indptr,
                                                          const int32_t *__restrict__ indices,
                                                          const T *__restrict__ data, int64_t n_rows, int32_t tile_log2,
                                                          int32_t n_terms, int32_t split,
                                                          const uint32_t *__restrict__ cnt /* [wgs][n_terms]: entries in earlier workgroups */,
                                                          const uint32_t *__restrict__ term_start,
                                                          const uint8_t *__restrict__ is_frequent, int32_t *out_rows,
                                                          T *out_vals, uint32_t *out_filt, float inv_norm_up, int32_t fold_log2) {
    extern __shared__ uint32_t cursor[];   // next free slot of (term k, this tile, this part); then the frequent-term bits
    uint32_t *freq_bits = cursor + n_terms;
    const int64_t t = blockIdx.x / split;
    const int32_t part = blockIdx.x % split;
    for (int k = threadIdx.x; k < (n_terms + 31) / 32; k += blockDim.x) freq_bits[k] = 0;
    const uint32_t *mine = cnt + (int64_t)blockIdx.x * n_terms;
    for (int k = threadIdx.x; k < n_terms; k += blockDim.x) cursor[k] = term_start[k] + mine[k];
    __syncthreads();
    if (out_filt)
        for (int k = threadIdx.x; k < n_terms; k += blockDim.x)
            if (is_frequent[k]) atomicOr(&freq_bits[k >> 5], 1u << (k & 31));
    __syncthreads();
    const int64_t part_rows = ((int64_t)1 << tile_log2) / split;
    const int64_t j0 = (t << tile_log2) + part * part_rows;
    int64_t j1 = j0 + part_rows;
    if (j1 > n_rows) j1 = n_rows;
    const int lane = threadIdx.x & 63, sub = lane & 15;
    (void)lane;
    const int64_t groups = blockDim.x >> 4;
    // every wave makes the same number of trips, so that the cross-lane sums below always run with all lanes
    const int64_t trips = (j1 - j0 + groups * SG_POST_ROWS - 1) / (groups * SG_POST_ROWS);
    for (int64_t it = 0; it < trips; ++it) {
        const int64_t jb = j0 + (it * groups + (threadIdx.x >> 4)) * SG_POST_ROWS;
        int64_t lo[SG_POST_ROWS], hi[SG_POST_ROWS];
#pragma unroll
        for (int r = 0; r < SG_POST_ROWS; ++r) {
            const bool valid = jb + r < j1;
            lo[r] = valid ? indptr[jb + r] : 0;
            hi[r] = valid ? indptr[jb + r + 1] : 0;
        }
        // the lane's first entry of every row: all loads before the first use (rows of up to sixteen entries are through
        // with these; the rest of a longer row goes entry by entry below)
        int32_t k0[SG_POST_ROWS];
        T v0[SG_POST_ROWS];
#pragma unroll
        for (int r = 0; r < SG_POST_ROWS; ++r) {
            const bool have = lo[r] + sub < hi[r];
            k0[r] = have ? indices[lo[r] + sub] : -1;
            v0[r] = have ? data[lo[r] + sub] : (T)0;
        }
        float fq[SG_POST_ROWS];
#pragma unroll
        for (int r = 0; r < SG_POST_ROWS; ++r) {
            fq[r] = 0.f;
#if defined(SG_K3_PROBE_NO_FQ)
            if (out_filt && n_rows < 0) {
#else
            if (out_filt) {
#endif
                // norm of the row's frequent part relative to norm_up, rounded up (the order of the additions is free: the
                // result is rounded up with a margin far above the rounding of a double sum)
                double f2 = 0.0;
                if (k0[r] >= 0 && ((freq_bits[k0[r] >> 5] >> (k0[r] & 31)) & 1u)) f2 = (double)v0[r] * (double)v0[r];
                for (int64_t p = lo[r] + sub + 16; p < hi[r]; p += 16) {
                    const int32_t k = indices[p];
                    if ((freq_bits[k >> 5] >> (k & 31)) & 1u) f2 += (double)data[p] * (double)data[p];
                }
#pragma unroll
                for (int d = 8; d > 0; d >>= 1) {
                    const uint64_t bits = (uint64_t)__double_as_longlong(f2);
                    const uint32_t l = (uint32_t)__shfl_xor((int)(uint32_t)bits, d, 64), h = (uint32_t)__shfl_xor((int)(uint32_t)(bits >> 32), d, 64);
                    f2 += __longlong_as_double((long long)(((uint64_t)h << 32) | l));
                }
                fq[r] = frequent_norm_ratio_of(f2, inv_norm_up);
            }
        }
#pragma unroll
        for (int r = 0; r < SG_POST_ROWS; ++r) {
            const uint32_t col = (uint32_t)(jb + r - (t << tile_log2));
            if (k0[r] >= 0) {
#if defined(SG_K3_PROBE_NO_ATOMIC)   // timing probes (wrong results): a build of the library per probe, A/B through SG_HIP_LIB
                const uint32_t pos = cursor[k0[r]] + (uint32_t)(threadIdx.x & 3);
#else
                const uint32_t pos = atomicAdd(&cursor[k0[r]], 1u);
#endif
#if defined(SG_K3_PROBE_NO_STORE)
                if (pos == 0xFFFFFFF0u)
#endif
                emit_posting<T>(out_rows, out_vals, out_filt, pos, col, v0[r], fq[r], tile_log2, inv_norm_up, (uint32_t)t, fold_log2);
            }
            for (int64_t p = lo[r] + sub + 16; p < hi[r]; p += 16) {
                const uint32_t pos = atomicAdd(&cursor[indices[p]], 1u);
                emit_posting<T>(out_rows, out_vals, out_filt, pos, col, data[p], fq[r], tile_log2, inv_norm_up, (uint32_t)t, fold_log2);
            }
        }
    }
}

// ---- Round 6: the filter postings STAGED in LDS and written cell by cell.
// postings_fill_lds stores every posting where its cursor says: 4 bytes to a line of its own -- the term-major order the
// multiply streams puts the postings of one row 18 300 lists apart --, and those stores were half the kernel (0.12 of 0.27 ms
// at 663 k, probe builds of round 4; 1.9 ms at 5 M).  But the entries of a (term, part) CELL are neighbours in the index, and
// most entries live in cells of many (frequent terms: a part of 1024 rows holds 4.3 entries per non-empty cell, 56 % of the
// entries in cells of eight or more).  So a workgroup works its part off in CHUNKS of rows whose postings fit the LDS beside
// the counters: (1) the chunk's entries are counted per term in packed 16-bit counters (and the rows' frequent-part norms
// computed); (2) a prefix sum turns the counts into the cells' places inside the chunk; (3) every posting is made and put
// at its place in LDS; (4) waves copy the staged run out, 64 consecutive staged postings a trip: lanes in the same cell write
// neighbouring words, a trip touches ~15 lines instead of 64.  Where a cell starts in the index is the workgroup's row of
// `cnt` (entries in earlier workgroups) + the term's start, advanced by the chunk's count -- the row is this workgroup's
// alone, and the tables the multiply reads are written from `cnt` BEFORE this kernel runs.  A chunk that does not fit the
// stage (rows far longer than the build's mean) is written posting by posting through the same cursors.
// Filter postings only: the exact kernel's postings, when they are wanted at all, keep postings_fill_lds.
__device__ __forceinline__ uint32_t lds_get16(const uint32_t *w, uint32_t k) { return (w[k >> 1] >> ((k & 1u) << 4)) & 0xffffu; }
template <typename T>
__global__ void __launch_bounds__(1024) postings_fill_staged(const int64_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                                                             const T *__restrict__ data, int64_t n_rows, int32_t tile_log2,
                                                             int32_t n_terms, int32_t split, int32_t chunk_rows, uint32_t stage_cap,
                                                             uint32_t *cnt /* [wgs][n_terms] */, const uint32_t *__restrict__ term_start,
                                                             const uint8_t *__restrict__ is_frequent, uint32_t *__restrict__ out_filt,
                                                             float inv_norm_up, int32_t fold_log2) {
    extern __shared__ uint32_t lds[];
    const uint32_t w16 = ((uint32_t)n_terms + 1u) >> 1;            // words of packed 16-bit counters
    uint32_t *lcnt = lds;
    uint32_t *freq_bits = lcnt + ((w16 + 3u) & ~3u);
    float *fqrow = reinterpret_cast<float *>(freq_bits + (((uint32_t)n_terms + 31u) >> 5));
    uint32_t *wave_tot = reinterpret_cast<uint32_t *>(fqrow + chunk_rows);
    uint32_t *stage = wave_tot + 32;
    const int64_t t = blockIdx.x / split;
    const int32_t part = blockIdx.x % split;
    uint32_t *mine = cnt + (int64_t)blockIdx.x * n_terms;
    for (uint32_t k = threadIdx.x; k < (((uint32_t)n_terms + 31u) >> 5); k += blockDim.x) freq_bits[k] = 0;
    __syncthreads();
    for (int k = threadIdx.x; k < n_terms; k += blockDim.x)
        if (is_frequent[k]) atomicOr(&freq_bits[k >> 5], 1u << (k & 31));
    __syncthreads();
    const int64_t part_rows = ((int64_t)1 << tile_log2) / split;
    const int64_t j0 = (t << tile_log2) + part * part_rows;
    int64_t j1 = j0 + part_rows;
    if (j1 > n_rows) j1 = n_rows;
    const int lane = threadIdx.x & 63, sub = lane & 15, wave = threadIdx.x >> 6;
    const int64_t groups = blockDim.x >> 4;
    const uint32_t last_k = (uint32_t)n_terms - 1u;
    // A workgroup whose part holds a chunk that does not fit the stage writes ALL its chunks posting by posting: its cursors
    // are then only ever touched by atomics, a staging workgroup's only by the plain loads and stores of the one thread that
    // owns a term -- never both (atomics are served by the L2, plain loads may come from the L1: mixing them per workgroup
    // needed loads and stores through the fabric, which doubled the kernel -- 0.53 instead of 0.26 ms at 663 k).
    bool any_big = false;
    for (int64_t c0 = j0; c0 < j1; c0 += chunk_rows) {
        const int64_t c1 = c0 + chunk_rows < j1 ? c0 + chunk_rows : j1;
        const int64_t e = indptr[c1] - indptr[c0];
        any_big = any_big || e > (int64_t)stage_cap || e > 65535;
    }
    for (int64_t c0 = j0; c0 < j1; c0 += chunk_rows) {
        const int64_t c1 = c0 + chunk_rows < j1 ? c0 + chunk_rows : j1;
        // every wave makes the same number of trips, so that the cross-lane sums below always run with all lanes
        const int64_t trips = (c1 - c0 + groups * SG_POST_ROWS - 1) / (groups * SG_POST_ROWS);
        if (any_big) {
            // (rows far longer than the build expected: posting by posting through the workgroup's cursors)
            for (int64_t it = 0; it < trips; ++it) {
                const int64_t jb = c0 + (it * groups + (threadIdx.x >> 4)) * SG_POST_ROWS;
#pragma unroll
                for (int r = 0; r < SG_POST_ROWS; ++r) {
                    const bool valid = jb + r < c1;
                    const int64_t lo = valid ? indptr[jb + r] : 0, hi = valid ? indptr[jb + r + 1] : 0;
                    double f2 = 0.0;
                    for (int64_t p = lo + sub; p < hi; p += 16) {
                        const int32_t k = indices[p];
                        if ((freq_bits[k >> 5] >> (k & 31)) & 1u) f2 += (double)data[p] * (double)data[p];
                    }
#pragma unroll
                    for (int d = 8; d > 0; d >>= 1) {
                        const uint64_t bits = (uint64_t)__double_as_longlong(f2);
                        const uint32_t l = (uint32_t)__shfl_xor((int)(uint32_t)bits, d, 64), h = (uint32_t)__shfl_xor((int)(uint32_t)(bits >> 32), d, 64);
                        f2 += __longlong_as_double((long long)(((uint64_t)h << 32) | l));
                    }
                    const float fq = frequent_norm_ratio_of(f2, inv_norm_up);
                    const uint32_t col = (uint32_t)(jb + r - (t << tile_log2));
                    for (int64_t p = lo + sub; p < hi; p += 16) {
                        const int32_t k = indices[p];
                        const uint32_t pos = term_start[k] + atomicAdd(&mine[k], 1u);
                        out_filt[pos] = filter_posting(col, (float)data[p], fq, tile_log2, inv_norm_up, (uint32_t)t, fold_log2);
                    }
                }
            }
            __syncthreads();
            continue;
        }
        for (uint32_t w = threadIdx.x; w < w16; w += blockDim.x) lcnt[w] = 0;
        __syncthreads();
        // ---- (1) count per term; the rows' frequent-part norms
        for (int64_t it = 0; it < trips; ++it) {
            const int64_t jb = c0 + (it * groups + (threadIdx.x >> 4)) * SG_POST_ROWS;
            int64_t lo[SG_POST_ROWS], hi[SG_POST_ROWS];
#pragma unroll
            for (int r = 0; r < SG_POST_ROWS; ++r) {
                const bool valid = jb + r < c1;
                lo[r] = valid ? indptr[jb + r] : 0;
                hi[r] = valid ? indptr[jb + r + 1] : 0;
            }
            int32_t k0[SG_POST_ROWS];
            T v0[SG_POST_ROWS];
#pragma unroll
            for (int r = 0; r < SG_POST_ROWS; ++r) {
                const bool have = lo[r] + sub < hi[r];
                k0[r] = have ? indices[lo[r] + sub] : -1;
                v0[r] = have ? data[lo[r] + sub] : (T)0;
            }
#pragma unroll
            for (int r = 0; r < SG_POST_ROWS; ++r) {
                // norm of the row's frequent part relative to norm_up, rounded up (the order of the additions is free: the
                // result is rounded up with a margin far above the rounding of a double sum)
                double f2 = 0.0;
                if (k0[r] >= 0) {
                    atomicAdd(&lcnt[(uint32_t)k0[r] >> 1], 1u << (((uint32_t)k0[r] & 1u) << 4));
                    if ((freq_bits[k0[r] >> 5] >> (k0[r] & 31)) & 1u) f2 = (double)v0[r] * (double)v0[r];
                }
                for (int64_t p = lo[r] + sub + 16; p < hi[r]; p += 16) {
                    const int32_t k = indices[p];
                    atomicAdd(&lcnt[(uint32_t)k >> 1], 1u << (((uint32_t)k & 1u) << 4));
                    if ((freq_bits[k >> 5] >> (k & 31)) & 1u) f2 += (double)data[p] * (double)data[p];
                }
#pragma unroll
                for (int d = 8; d > 0; d >>= 1) {
                    const uint64_t bits = (uint64_t)__double_as_longlong(f2);
                    const uint32_t l = (uint32_t)__shfl_xor((int)(uint32_t)bits, d, 64), h = (uint32_t)__shfl_xor((int)(uint32_t)(bits >> 32), d, 64);
                    f2 += __longlong_as_double((long long)(((uint64_t)h << 32) | l));
                }
                if (sub == 0 && jb + r < c1) fqrow[jb + r - c0] = frequent_norm_ratio_of(f2, inv_norm_up);
            }
        }
        __syncthreads();
        // ---- (2) counts -> where every term's cell starts inside the chunk (a thread takes a run of words)
        {
            const uint32_t per = (w16 + blockDim.x - 1u) / blockDim.x;
            const uint32_t wa = threadIdx.x * per, wb = wa + per < w16 ? wa + per : w16;
            uint32_t sum = 0;
            for (uint32_t w = wa; w < wb; ++w) sum += (lcnt[w] & 0xffffu) + (lcnt[w] >> 16);
            uint32_t tot;
            uint32_t run = block_exclusive_scan<uint32_t>(sum, wave_tot, &tot);
            for (uint32_t w = wa; w < wb; ++w) {
                const uint32_t c = lcnt[w];
                const uint32_t a = run;
                run += c & 0xffffu;
                const uint32_t b = run;
                run += c >> 16;
                lcnt[w] = a | (b << 16);
            }
        }
        __syncthreads();
        // ---- (3) the postings, each at its place in LDS (the returning add leaves every counter at its cell's END)
        for (int64_t it = 0; it < trips; ++it) {
            const int64_t jb = c0 + (it * groups + (threadIdx.x >> 4)) * SG_POST_ROWS;
            int64_t lo[SG_POST_ROWS], hi[SG_POST_ROWS];
#pragma unroll
            for (int r = 0; r < SG_POST_ROWS; ++r) {
                const bool valid = jb + r < c1;
                lo[r] = valid ? indptr[jb + r] : 0;
                hi[r] = valid ? indptr[jb + r + 1] : 0;
            }
            int32_t k0[SG_POST_ROWS];
            T v0[SG_POST_ROWS];
#pragma unroll
            for (int r = 0; r < SG_POST_ROWS; ++r) {
                const bool have = lo[r] + sub < hi[r];
                k0[r] = have ? indices[lo[r] + sub] : -1;
                v0[r] = have ? data[lo[r] + sub] : (T)0;
            }
#pragma unroll
            for (int r = 0; r < SG_POST_ROWS; ++r) {
                const float fq = jb + r < c1 ? fqrow[jb + r - c0] : 0.f;
                const uint32_t col = (uint32_t)(jb + r - (t << tile_log2));
                if (k0[r] >= 0) {
                    const uint32_t sh = ((uint32_t)k0[r] & 1u) << 4;
                    const uint32_t pos = (atomicAdd(&lcnt[(uint32_t)k0[r] >> 1], 1u << sh) >> sh) & 0xffffu;
                    stage[pos] = filter_posting(col, (float)v0[r], fq, tile_log2, inv_norm_up, (uint32_t)t, fold_log2);
                }
                for (int64_t p = lo[r] + sub + 16; p < hi[r]; p += 16) {
                    const uint32_t k = (uint32_t)indices[p];
                    const uint32_t sh = (k & 1u) << 4;
                    const uint32_t pos = (atomicAdd(&lcnt[k >> 1], 1u << sh) >> sh) & 0xffffu;
                    stage[pos] = filter_posting(col, (float)data[p], fq, tile_log2, inv_norm_up, (uint32_t)t, fold_log2);
                }
            }
        }
        __syncthreads();
        // ---- (4) out: a wave takes 64 terms at a time, whose cells are one run of the stage.  Where the cells start in the
        // index (the workgroup's cursors: a round trip to the L2) is fetched for EIGHT such groups before the first is copied:
        // one group after the other, the kernel waited for that load eighteen times a chunk (0.26 of its 0.27 ms at 663 k).
        {
            constexpr int FB = 8;
            const uint32_t n_tg = ((uint32_t)n_terms + 63u) >> 6, waves = (uint32_t)(blockDim.x >> 6);
            for (uint32_t tg0 = (uint32_t)wave; tg0 < n_tg; tg0 += waves * FB) {
                uint32_t end_k[FB], start_k[FB], before[FB], G[FB];
                bool mine_has[FB];
#pragma unroll
                for (int b = 0; b < FB; ++b) {
                    const uint32_t tg = tg0 + (uint32_t)b * waves;
                    const uint32_t k = tg * 64u + (uint32_t)lane;
                    const bool live = tg < n_tg;
                    const uint32_t kc = k < last_k ? k : last_k;                   // (lanes past the last term: empty cells at the end)
                    end_k[b] = live ? lds_get16(lcnt, kc) : 0u;
                    start_k[b] = (!live || k == 0u) ? 0u : lds_get16(lcnt, (k - 1u) < last_k ? k - 1u : last_k);
                    mine_has[b] = live && k <= last_k && end_k[b] > start_k[b];
                }
#pragma unroll
                for (int b = 0; b < FB; ++b) {
                    const uint32_t k = (tg0 + (uint32_t)b * waves) * 64u + (uint32_t)lane;
                    before[b] = mine_has[b] ? mine[k] : 0u;
                    G[b] = mine_has[b] ? term_start[k] : 0u;
                }
#pragma unroll
                for (int b = 0; b < FB; ++b) {
                    const uint32_t tg = tg0 + (uint32_t)b * waves;
                    if (tg >= n_tg) break;
                    const uint32_t k = tg * 64u + (uint32_t)lane;
                    G[b] += before[b];
                    const uint32_t S = (uint32_t)__builtin_amdgcn_readlane((int)start_k[b], 0);
                    const uint32_t Eg = (uint32_t)__builtin_amdgcn_readlane((int)end_k[b], 63);
                    for (uint32_t e0 = S; e0 < Eg; e0 += 64u) {
                        const uint32_t e = e0 + (uint32_t)lane;
                        // the cell of staged posting e: the first of the 64 terms whose end lies behind e
                        uint32_t lo = 0u, hi = 63u;
#pragma unroll
                        for (int step = 0; step < 6; ++step) {
                            const uint32_t mid = (lo + hi) >> 1;
                            const uint32_t km = tg * 64u + mid;
                            const bool right = lds_get16(lcnt, km < last_k ? km : last_k) > e;
                            hi = right ? mid : hi;
                            lo = right ? lo : mid + 1u;
                        }
                        const uint32_t tsel = lo < 63u ? lo : 63u;
                        const uint32_t kt = tg * 64u + tsel;
                        const uint32_t start_t = kt == 0u ? 0u : lds_get16(lcnt, (kt - 1u) < last_k ? kt - 1u : last_k);
                        const uint32_t G_t = (uint32_t)__shfl((int)G[b], (int)tsel, 64);
                        if (e < Eg) out_filt[G_t + (e - start_t)] = stage[e];
                    }
                    if (mine_has[b]) mine[k] = before[b] + (end_k[b] - start_k[b]);
                }
            }
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256) term_len_kernel(const uint32_t *__restrict__ seg, int64_t n_terms, int32_t n_tiles,
                                                       uint32_t *__restrict__ term_len) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n_terms) term_len[k] = seg[(k + 1) * n_tiles] - seg[k * n_tiles];
}

// Packed copy of B's rows for the pruned multiply (one 16-byte load = two f32 entries or one f64 entry): a thread per
// entry, and a thread per row for its {32-bit row pointer, the row's own index} -- the index rides with the pointer
// because the multiply needs both for every pair it scores: as a table of its own (orig_of[j]) it cost a cache line per
// pair, 4 GB of the kernel's 55 GB at 663 k.
template <typename T>
__global__ void __launch_bounds__(256) fwd_pack(const int64_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                                                const T *__restrict__ data, int64_t n_rows, int64_t nnz,
                                                const uint32_t *__restrict__ orig_of /* position -> row; null: identity */,
                                                uint32_t *__restrict__ fwd_ptr /* uint2 per row: {pointer, row} */, void *__restrict__ fwd) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t base = indptr[0];
    if (i <= n_rows)
        reinterpret_cast<uint2 *>(fwd_ptr)[i] = make_uint2((uint32_t)(indptr[i] - base), i < n_rows ? (orig_of ? orig_of[i] : (uint32_t)i) : 0u);
    if (i >= nnz || fwd == nullptr) return;   // (fwd null: the row blocks carry the rows, only the pointer table is made)
    const int64_t p = base + i;
    if (sizeof(T) == 4) {
        reinterpret_cast<int2 *>(fwd)[i] = make_int2(indices[p], __float_as_int((float)data[p]));
    } else {
        const long long bits = __double_as_longlong((double)data[p]);
        reinterpret_cast<int4 *>(fwd)[i] = make_int4(indices[p], 0, (int)(bits & 0xffffffffll), (int)(bits >> 32));
    }
}

// Row blocks for the exact scoring of the pruned multiply (SgScoreCtx::blk): one thread per 16-byte chunk of the
// 128-byte lines a row uses.
template <typename T>
__global__ void __launch_bounds__(256) row_blocks_kernel(const int64_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                                                         const T *__restrict__ data, int64_t n_rows,
                                                         const uint32_t *__restrict__ orig_of /* position -> row; null: identity */,
                                                         uint32_t blk_bytes, void *__restrict__ blk) {
    constexpr int ES = sizeof(T) == 4 ? 8 : 16;    // bytes of an entry
    const int64_t chunks = blk_bytes / 16;
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t p = g / chunks;
    const int32_t c = (int32_t)(g - p * chunks);
    if (p >= n_rows) return;
    const int64_t lo = indptr[p];
    const int32_t nnz = (int32_t)(indptr[p + 1] - lo);
    const int32_t used = (((nnz + 1) * ES + 127) / 128) * 128;     // bytes of the lines the row uses
    if (c * 16 >= used) return;
    uint4 w = make_uint4(0u, 0u, 0u, 0u);
    if (sizeof(T) == 4) {   // two entries per chunk: 2c, 2c + 1 (entry 0 = header)
        const int32_t e0 = 2 * c, e1 = 2 * c + 1;
        if (e0 == 0) {
            w.x = orig_of ? orig_of[p] : (uint32_t)p;
            w.y = (uint32_t)nnz;
        } else if (e0 <= nnz) {
            w.x = (uint32_t)indices[lo + e0 - 1];
            w.y = __float_as_uint((float)data[lo + e0 - 1]);
        }
        if (e1 <= nnz) {
            w.z = (uint32_t)indices[lo + e1 - 1];
            w.w = __float_as_uint((float)data[lo + e1 - 1]);
        }
    } else {                // one entry per chunk
        if (c == 0) {
            w.x = orig_of ? orig_of[p] : (uint32_t)p;
            w.y = (uint32_t)nnz;
        } else if (c <= nnz) {
            const long long bits = __double_as_longlong((double)data[lo + c - 1]);
            w.x = (uint32_t)indices[lo + c - 1];
            w.z = (uint32_t)(bits & 0xffffffffll);
            w.w = (uint32_t)(bits >> 32);
        }
    }
    reinterpret_cast<uint4 *>(reinterpret_cast<char *>(blk) + (size_t)p * blk_bytes)[c] = w;
}

// (path with global counters) the two padded tables of segment ends from the term-major segment table: a thread per term
__global__ void __launch_bounds__(256) ends_from_seg_kernel(const uint32_t *__restrict__ seg, int64_t n_terms, int32_t n_tiles,
                                                            uint32_t *__restrict__ ends, int32_t nt_pad, uint32_t *__restrict__ ends8,
                                                            int32_t nv_pad, int32_t fold_log2) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k > n_terms) return;
    if (ends)
        for (int32_t t = 0; t < nt_pad; ++t)
            ends[k * nt_pad + t] = k == n_terms ? 0u : seg[k * n_tiles + (t < n_tiles ? t : n_tiles - 1) + 1] << 2;
    if (ends8)
        for (int32_t v = 0; v < nv_pad; ++v) {
            int64_t t = (int64_t)(v + 1) << fold_log2;   // first tile past super-tile v
            if (t > n_tiles) t = n_tiles;
            ends8[k * nv_pad + v] = k == n_terms ? 0u : seg[k * n_tiles + t] << 2;
        }
}

__global__ void __launch_bounds__(256) null_postings_kernel(uint32_t *__restrict__ slack) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;   // 512 entries
    slack[i] = sg_null_posting(i);
}

// ---- position space: a fixed permutation of the right-hand rows (see sg_postings in sg_internal.h)
// pos_of[j] = j * M mod n with gcd(M, n) = 1, M ~ 0.618 n: neighbours land 0.618 n apart and any run of rows spreads
// evenly over the positions (a low-discrepancy sequence) -- what a sorted list needs, and harmless on any other.
// Round 6: no launch of its own -- the prefix sum over the rows' lengths in position order (the row pointers of the matrix in
// position order) computes the permutation as it loads: position p holds row g = p * M^-1 mod n.
struct PermutedLenLoad {
    uint64_t n, minv, magic;       // magic = floor((2^64 - 1) / n) + 1: p * minv mod n by one multiply-high (a 64-bit `%` is ~100 instructions)
    const int32_t *len_of_row;     // lengths by row (groups: SgCollapse::d_rep_len); null: from the row pointers
    const int64_t *indptr;
    uint32_t *pos_of, *orig_of;
    __device__ __forceinline__ int64_t operator()(int64_t p) const {
        const uint64_t x = (uint64_t)p * minv;                    // < 2^62
        const uint64_t q = __umul64hi(x, magic);                  // floor(x / n) or one more
        int64_t rem = (int64_t)(x - q * n);
        if (rem < 0) rem += (int64_t)n;
        const uint32_t g = (uint32_t)rem;
        orig_of[p] = g;
        pos_of[g] = (uint32_t)p;
        return len_of_row ? (int64_t)len_of_row[g] : indptr[g + 1] - indptr[g];
    }
};

// ---- the 8-bit copies of the rows for the pruned multiply's SECOND filter (SgScoreCtx::q8, sg_internal.h; round 5)
// One 16-byte unit of a row's record: unit 0 = {first packed entry, the row's own index, entries | flag, 0}, unit u >= 1 =
// entries 4u - 4 .. 4u - 1; an entry is (term << 8) | bq with bq = ceil(value / norm_up * 255) -- UP: the factor 1.000002
// pays for the float roundings of the product (four of 2^-24), the cut at 255 is sound because no value exceeds its
// row's norm.  Units a row does not reach are neither written nor read.
__device__ __forceinline__ uint32_t sg_q8_units(int64_t nnz) {
    return nnz > (int64_t)SG_Q8_MAX_ENTRIES ? 1u : (uint32_t)((nnz + 7) >> 2);   // (the header + four entries a unit)
}
template <typename T>
__device__ __forceinline__ void q8_write_unit(uint4 *__restrict__ rec, uint32_t u, const int32_t *__restrict__ indices,
                                              const T *__restrict__ data, int64_t src, int64_t nnz, uint32_t first_packed,
                                              uint32_t name, float inv_norm) {
    auto entry = [&](int64_t e) -> uint32_t {
        if (e >= nnz) return 0u;
        const float q = ceilf((float)data[src + e] * inv_norm * 255.0f * 1.000002f);
        const uint32_t bq = q >= 255.0f ? 255u : (q >= 1.0f ? (uint32_t)q : 1u);
        return ((uint32_t)indices[src + e] << 8) | bq;
    };
    uint4 w;
    if (u == 0) {
        w = make_uint4(first_packed, name, (uint32_t)nnz | (nnz <= (int64_t)SG_Q8_MAX_ENTRIES ? 0u : 0x80000000u), 0u);
    } else {
        const int64_t e0 = 4 * (int64_t)u - 4;
        w = make_uint4(entry(e0), entry(e0 + 1), entry(e0 + 2), entry(e0 + 3));
    }
    rec[u] = w;
}

// (the index without a copy of the rows in position order -- SG_PERMUTE=0, few rows: sixteen lanes per row)
template <typename T>
__global__ void __launch_bounds__(256) q8_pack_kernel(const int64_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                                                      const T *__restrict__ data, int64_t n_rows,
                                                      const uint32_t *__restrict__ orig_of /* position -> row; null: identity */,
                                                      uint4 *__restrict__ q8, float inv_norm) {
    const int64_t p = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const uint32_t sub = threadIdx.x & 15;
    if (p >= n_rows) return;
    const int64_t base = indptr[0], src = indptr[p], n = indptr[p + 1] - src;
    if (sub < sg_q8_units(n))
        q8_write_unit<T>(q8 + p * (SG_Q8_STRIDE / 16), sub, indices, data, src, n, (uint32_t)(src - base),
                         orig_of ? orig_of[p] : (uint32_t)p, inv_norm);
}

// ONE read of the source rows for every copy of them the index keeps (round 4).  Position p of the index holds row
// g = orig_of[p] of the matrix it is built over; with groups of identical rows that matrix is the representatives' -- row g
// = row rep_rows[g] of the caller's matrix -- and the copies are
//   * the matrix in position order (what the index and the self-join form read),
//   * its packed rows {term, value} + {pointer, row} per position (the exact scoring of the pruned multiply),
//   * the 8-bit records of the second filter.
// Round 6: (a) the representatives' matrix in GROUP order is no longer written here -- nothing on the self-join's way reads
// it (sg_csr_ensure_rows writes it for whoever does): 88 of 420 MB at 663 k; (b) where a row starts in the source and how
// long it is come from two arrays the grouping leaves per group (rep_start, rep_len): the chain position -> group ->
// representative -> row pointers -> entries was four dependent loads deep, now two; (c) TWO rows per sixteen lanes with all
// loads of a stage issued before the first is used, and a row's first 32 entries in flight together: the kernel ran at the
// latency of its chain (0.21 ms for 0.4 GB); (d) the 8-bit records are put together from the registers that hold the
// entries (four shuffles a round), not by reading the row a second time.
#ifndef SG_GATHER_ROWS     // (A/B knob of the build: scripts/build_variant.sh)
#define SG_GATHER_ROWS 2
#endif
template <typename T>
__device__ __forceinline__ uint32_t q8_entry_of(int32_t k, T v, bool have, float inv_norm) {
    if (!have) return 0u;
    const float q = ceilf((float)v * inv_norm * 255.0f * 1.000002f);
    const uint32_t bq = q >= 255.0f ? 255u : (q >= 1.0f ? (uint32_t)q : 1u);
    return ((uint32_t)k << 8) | bq;
}
// round r of a row (entries 16 r .. 16 r + 15, one per lane: wq) -> units 4 r + 1 .. 4 r + 4 of its record, written by the
// lanes of those numbers (unit u = entries 4 u - 4 .. 4 u - 1)
__device__ __forceinline__ void q8_write_round(uint4 *__restrict__ rec, int r, uint32_t wq, int sub, uint32_t units) {
    const int q = (sub - 4 * r - 1) & 3;
    const int base = (int)(threadIdx.x & 48u) + 4 * q;
    const uint32_t x0 = (uint32_t)__shfl((int)wq, base, 64), x1 = (uint32_t)__shfl((int)wq, base + 1, 64);
    const uint32_t x2 = (uint32_t)__shfl((int)wq, base + 2, 64), x3 = (uint32_t)__shfl((int)wq, base + 3, 64);
    if (sub >= 4 * r + 1 && sub <= 4 * r + 4 && (uint32_t)sub < units) rec[sub] = make_uint4(x0, x1, x2, x3);
}
template <typename T>
__device__ __forceinline__ void store_packed(void *__restrict__ fwd, int64_t at, int32_t k, T v) {
    if (sizeof(T) == 4) {
        reinterpret_cast<int2 *>(fwd)[at] = make_int2(k, __float_as_int((float)v));
    } else {
        const long long bits = __double_as_longlong((double)v);
        reinterpret_cast<int4 *>(fwd)[at] = make_int4(k, 0, (int)(bits & 0xffffffffll), (int)(bits >> 32));
    }
}
template <typename T>
__global__ void __launch_bounds__(256) gather_rows_kernel(const int64_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                                                          const T *__restrict__ data, int64_t n_rows,
                                                          const uint32_t *__restrict__ orig_of,
                                                          const uint32_t *__restrict__ rep_rows /* null: the source IS the matrix */,
                                                          const int64_t *__restrict__ rep_start /* null: from rep_rows / the row pointers */,
                                                          const int32_t *__restrict__ rep_len,
                                                          const int64_t *__restrict__ perm_ptr, int32_t *__restrict__ perm_indices,
                                                          T *__restrict__ perm_data,
                                                          uint32_t *__restrict__ fwd_ptr /* null: no packed rows */, void *__restrict__ fwd,
                                                          uint4 *__restrict__ q8 /* null: no 8-bit copies */, float inv_norm) {
    const int64_t grp = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const int sub = threadIdx.x & 15;
    const int64_t p0 = grp * SG_GATHER_ROWS;
    if (p0 > n_rows) return;
    bool valid[SG_GATHER_ROWS];
    uint32_t g[SG_GATHER_ROWS];
    int64_t src[SG_GATHER_ROWS], dst[SG_GATHER_ROWS];
    int32_t n[SG_GATHER_ROWS];
#pragma unroll
    for (int i = 0; i < SG_GATHER_ROWS; ++i) {
        valid[i] = p0 + i < n_rows;
        g[i] = valid[i] ? orig_of[p0 + i] : 0u;
        dst[i] = p0 + i <= n_rows ? perm_ptr[p0 + i] : 0;
    }
    if (rep_start) {
#pragma unroll
        for (int i = 0; i < SG_GATHER_ROWS; ++i) {
            src[i] = valid[i] ? rep_start[g[i]] : 0;
            n[i] = valid[i] ? rep_len[g[i]] : 0;
        }
    } else {
        int64_t j[SG_GATHER_ROWS];
#pragma unroll
        for (int i = 0; i < SG_GATHER_ROWS; ++i) j[i] = (valid[i] && rep_rows) ? (int64_t)rep_rows[g[i]] : (int64_t)g[i];
#pragma unroll
        for (int i = 0; i < SG_GATHER_ROWS; ++i) {
            src[i] = valid[i] ? indptr[j[i]] : 0;
            n[i] = valid[i] ? (int32_t)(indptr[j[i] + 1] - src[i]) : 0;
        }
    }
    // a row's first two rounds of entries: all loads before the first use
    int32_t k[SG_GATHER_ROWS][2];
    T v[SG_GATHER_ROWS][2];
#pragma unroll
    for (int i = 0; i < SG_GATHER_ROWS; ++i)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const bool have = sub + 16 * r < n[i];
            k[i][r] = have ? indices[src[i] + sub + 16 * r] : 0;
            v[i][r] = have ? data[src[i] + sub + 16 * r] : (T)0;
        }
#pragma unroll
    for (int i = 0; i < SG_GATHER_ROWS; ++i) {
        const int64_t p = p0 + i;
        if (p == n_rows) {     // the sentinel position behind the last row
            if (fwd_ptr && sub == 0) reinterpret_cast<uint2 *>(fwd_ptr)[p] = make_uint2((uint32_t)dst[i], 0u);
            // the exact scoring reads packed rows in rounds of eight entries and multiplies what lies past a row's end by a = 0:
            // the pad behind the LAST row must hold finite values (0 * NaN would poison that row's score)
            if (fwd && sub < 8) store_packed<T>(fwd, dst[i] + sub, 0, (T)0);
        }
        if (!valid[i]) continue;
        const uint32_t units = q8 ? sg_q8_units(n[i]) : 0u;
        uint4 *rec = q8 ? q8 + p * (SG_Q8_STRIDE / 16) : nullptr;
        if (sub == 0) {
            if (fwd_ptr) reinterpret_cast<uint2 *>(fwd_ptr)[p] = make_uint2((uint32_t)dst[i], g[i]);
            if (q8) rec[0] = make_uint4((uint32_t)dst[i], g[i], (uint32_t)n[i] | (n[i] <= (int32_t)SG_Q8_MAX_ENTRIES ? 0u : 0x80000000u), 0u);
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int32_t e = sub + 16 * r;
            const bool have = e < n[i];
            if (have) {
                perm_indices[dst[i] + e] = k[i][r];
                perm_data[dst[i] + e] = v[i][r];
                if (fwd) store_packed<T>(fwd, dst[i] + e, k[i][r], v[i][r]);
            }
            if (units > 1u && 16 * r < n[i]) q8_write_round(rec, r, q8_entry_of<T>(k[i][r], v[i][r], have, inv_norm), sub, units);
        }
        for (int r = 2; 16 * r < n[i]; ++r) {     // rows beyond 32 entries
            const int32_t e = sub + 16 * r;
            const bool have = e < n[i];
            const int32_t kk = have ? indices[src[i] + e] : 0;
            const T vv = have ? data[src[i] + e] : (T)0;
            if (have) {
                perm_indices[dst[i] + e] = kk;
                perm_data[dst[i] + e] = vv;
                if (fwd) store_packed<T>(fwd, dst[i] + e, kk, vv);
            }
            if (units > 1u && r < 4) q8_write_round(rec, r, q8_entry_of<T>(kk, vv, have, inv_norm), sub, units);
        }
    }
}

__global__ void score_ctx_kernel(SgScoreCtx v, SgScoreCtx *out) {
    out[0] = v;
    v.q8 = nullptr;          // [1]: the same without the second filter (the multiply chooses per call: sg_q8_applies)
    v.q8_scale = 0.f;
    out[1] = v;
}

static uint64_t gcd_u64(uint64_t a, uint64_t b) {
    while (b) {
        const uint64_t t = a % b;
        a = b;
        b = t;
    }
    return a;
}

// B with its rows in position order + the two tables; *out_perm stays null when the permutation is off or pointless
static int build_permuted(sg_ctx *ctx, const sg_csr *B, int64_t tile_cols, sg_csr **out_perm, uint32_t **out_orig_of,
                          uint32_t **out_pos_of, SgCollapse *pending /* B = pending->unique, its rows not written yet; or null */,
                          uint32_t *fwd_ptr, void *fwd /* packed rows to write along (null: none) */, bool *fwd_done,
                          void *q8 = nullptr /* 8-bit copies to write along */, float inv_norm = 0.f) {
    *fwd_done = false;
    *out_perm = nullptr;
    *out_orig_of = *out_pos_of = nullptr;
    const char *e = ctx->opt("SG_PERMUTE");
    if ((e && e[0] == '0') || B->n_rows <= 2 * tile_cols || B->n_rows >= ((int64_t)1 << 31) || B->nnz <= 0) return SG_OK;
    const uint64_t n = (uint64_t)B->n_rows;
    uint64_t mult = (uint64_t)(0.6180339887498949 * (double)n) | 1ull;
    while (gcd_u64(mult, n) != 1) mult += 2;
    mult %= n;
    // position p holds row p * mult^-1 mod n (extended Euclid; n < 2^31)
    uint64_t minv = 0;
    {
        long long t0 = 0, t1 = 1, r0 = (long long)n, r1 = (long long)mult;
        while (r1 != 0) {
            const long long q = r0 / r1, t2 = t0 - q * t1, r2 = r0 - q * r1;
            t0 = t1;
            t1 = t2;
            r0 = r1;
            r1 = r2;
        }
        minv = (uint64_t)(t0 < 0 ? t0 + (long long)n : t0);
    }
    uint32_t *orig_of = nullptr, *pos_of = nullptr, *len_by_pos = nullptr;
    int64_t *ptr = nullptr;
    int32_t *idx = nullptr;
    void *val = nullptr;
    const size_t vs = B->dtype == SG_F64 ? 8 : 4;
    // the rows of B itself, or -- B the representatives' matrix of `pending`, not written -- of the matrix they come from
    const bool from_groups = pending && pending->pending_src;
    const bool by_group = from_groups && pending->d_rep_len != nullptr;   // (table path: start and length per group; else B's row pointers are there)
    int st = sg_alloc(ctx, (size_t)n + 1, &orig_of);
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)n + 1, &pos_of);
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)n + 2, &ptr);
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)B->nnz + 64, &idx);
    if (st == SG_OK) st = ctx->alloc(((size_t)B->nnz + 64) * vs, &val);
    if (st == SG_OK)
        st = sg_scan_launch<int64_t>(ctx, PermutedLenLoad{n, minv, (uint64_t)(~0ull / n) + 1ull, by_group ? pending->d_rep_len : (const int32_t *)nullptr, B->d_indptr, pos_of, orig_of},
                                     SgScanStoreArray<int64_t>{ptr}, (int64_t)n, ptr + n);
    if (st == SG_OK) {
        const unsigned grid = (unsigned)((((n + 1 + SG_GATHER_ROWS - 1) / SG_GATHER_ROWS) * 16 + 255) / 256);
        const sg_csr *src = from_groups ? pending->pending_src : B;
        const uint32_t *rep_rows = from_groups ? pending->d_rep_rows : nullptr;
        const int64_t *rep_start = by_group ? pending->d_rep_start : nullptr;
        const int32_t *rep_len = by_group ? pending->d_rep_len : nullptr;
        if (B->dtype == SG_F64)
            hipLaunchKernelGGL(gather_rows_kernel<double>, dim3(grid), dim3(256), 0, ctx->stream, src->d_indptr, src->d_indices,
                               (const double *)src->d_data, B->n_rows, (const uint32_t *)orig_of, rep_rows, rep_start, rep_len,
                               (const int64_t *)ptr, idx, (double *)val, fwd_ptr, fwd, (uint4 *)(fwd_ptr ? q8 : nullptr), inv_norm);
        else
            hipLaunchKernelGGL(gather_rows_kernel<float>, dim3(grid), dim3(256), 0, ctx->stream, src->d_indptr, src->d_indices,
                               (const float *)src->d_data, B->n_rows, (const uint32_t *)orig_of, rep_rows, rep_start, rep_len,
                               (const int64_t *)ptr, idx, (float *)val, fwd_ptr, fwd, (uint4 *)(fwd_ptr ? q8 : nullptr), inv_norm);
        if (hipGetLastError() != hipSuccess) st = SG_ERR_HIP;
        if (st == SG_OK) *fwd_done = fwd_ptr != nullptr;     // (the representatives' matrix in group order stays pending)
    }
    ctx->release(len_by_pos);
    sg_csr *m = st == SG_OK ? new (std::nothrow) sg_csr() : nullptr;
    if (st == SG_OK && !m) st = SG_ERR_OOM;
    if (st != SG_OK) {
        ctx->release(orig_of);
        ctx->release(pos_of);
        ctx->release(ptr);
        ctx->release(idx);
        ctx->release(val);
        return st;
    }
    m->ctx = ctx;
    m->n_rows = B->n_rows;
    m->n_cols = B->n_cols;
    m->nnz = B->nnz;
    m->dtype = B->dtype;
    m->d_indptr = ptr;
    m->d_indices = idx;
    m->d_data = val;
    m->owned = true;
    m->props_state = B->props_state;         // same rows: same properties
    m->props_max_norm2 = B->props_max_norm2;
    m->props_max_nnz = B->props_max_nnz;
    *out_perm = m;
    *out_orig_of = orig_of;
    *out_pos_of = pos_of;
    return SG_OK;
}

extern "C" int sg_postings_build(sg_ctx *ctx, const sg_csr *B, int32_t tile_cols, sg_postings **out) {
    return sg_postings_build_flags(ctx, B, tile_cols, 0, out);
}

// internal flags of sg_postings_build_flags (above the public ones)
#define SG_POSTINGS_NO_COLLAPSE (1 << 8)   // index every row (the collapse wrapper's own inner call; the on-demand plain index)
#define SG_POSTINGS_INNER (1 << 9)         // called by the collapse wrapper: the wrapper's timer covers the build
// (SG_POSTINGS_EXACT_ONLY, sg_internal.h: nothing of the pruned multiply's -- the exact kernel's own tile and postings)

// the groups whose representatives' rows the inner build of sg_postings_build_flags is to write (handed from the outer call
// to the inner one of the same thread)
static thread_local SgCollapse *tl_pending_groups = nullptr;

extern "C" int sg_postings_build_flags(sg_ctx *ctx, const sg_csr *B_in, int32_t tile_cols, int32_t flags, sg_postings **out) {
    SG_REQUIRE(ctx && B_in && out, "null argument");
    SgCollapse *pending = tl_pending_groups;
    tl_pending_groups = nullptr;
    if (!(flags & SG_POSTINGS_NO_COLLAPSE)) {
        // identical rows (identical strings): one representative per group is indexed (sg_collapse.hip)
        bool cl = false;
        float n2 = 0.f;
        SG_TRY(sg_csr_props(ctx, B_in, &cl, &n2));
        SgCollapse *col = nullptr;
        SgTimer timer(ctx, SG_K_POSTINGS);   // grouping + the index over the representatives
        if (cl) SG_TRY(sg_collapse_build(ctx, B_in, &col, /*left_side=*/false, /*defer_rows=*/true));
        if (col) {
            sg_postings *inner = nullptr;
            tl_pending_groups = col;      // (the inner build writes the representatives' rows with its own copies of them)
            const int st = sg_postings_build_flags(ctx, col->unique, tile_cols, flags | SG_POSTINGS_NO_COLLAPSE | SG_POSTINGS_INNER, &inner);
            tl_pending_groups = nullptr;
            if (st != SG_OK) {
                sg_collapse_free(col);
                return st;
            }
            inner->collapse = col;
            inner->caller_b_copy = *B_in;
            inner->caller_b_copy.owned = false;
            inner->caller_b_copy.d_props_words = nullptr;
            inner->caller_b_copy.left_groups = nullptr;
            inner->caller_b_copy.left_state = 0;
            inner->caller_b = &inner->caller_b_copy;
            inner->n_right_caller = B_in->n_rows;
            inner->build_tile_cols = tile_cols;
            inner->build_flags = flags;
            *out = inner;
            return SG_OK;
        }
        flags |= SG_POSTINGS_INNER;          // (no groups: this call goes on under the timer above)
        const int st = sg_postings_build_flags(ctx, B_in, tile_cols, flags | SG_POSTINGS_NO_COLLAPSE, out);
        if (st == SG_OK) {
            (*out)->build_tile_cols = tile_cols;
            (*out)->build_flags = flags & 0xff;
        }
        return st;
    }
    const sg_csr *B = B_in;
    // cosine-like right-hand sides (non-negative, sorted rows, norms <= 1: TF-IDF) take the pruned multiply,
    // whose 16-bit accumulators make a 4096-column tile 8 KiB; everything else the exact kernel with 8 KiB
    // of float / double accumulators per wave
    bool cosine_like = false;
    float max_norm2 = 0.f;
    SG_TRY(sg_csr_props(ctx, B, &cosine_like, &max_norm2));
    const char *pr = ctx->opt("SG_PRUNE");
    const bool want_pruned = cosine_like && !(pr && pr[0] == '0') && !(flags & SG_POSTINGS_EXACT_ONLY);
    if (tile_cols == 0) {
        tile_cols = B->dtype == SG_F64 ? 1024 : 2048;
        if (want_pruned) {
            tile_cols = 4096;
            if (flags & SG_POSTINGS_TILE_FORM) tile_cols = 2048;   // (the tile-by-tile form's own index: sg_spgemm_topn.hip, "which form")
            else if (const char *v = ctx->opt("SG_PRUNE_TILE")) tile_cols = atoi(v) == 13 ? 8192 : (atoi(v) == 11 ? 2048 : 4096);
        }
    }
    SG_REQUIRE(tile_cols >= 256 && tile_cols <= 32768 && (tile_cols & (tile_cols - 1)) == 0,
               "tile_cols must be a power of two in [256, 32768]");
    int64_t max_entries = (int64_t)1 << 29;   // the multiply addresses postings with 32-bit BYTE offsets (8 B entries)
    if (const char *v = ctx->opt("SG_MAX_POSTINGS")) {   // test hook: force the right-hand split at small sizes
        const long long o = atoll(v);
        if (o > 0 && o < max_entries) max_entries = o;
    }
    if (B->nnz + 64 >= max_entries) {
        sg_set_error("right-hand matrix has %lld non-zeros; one postings block holds < 2^29 (use more right-hand blocks)",
                     (long long)B->nnz);
        return SG_ERR_OVERFLOW;
    }
    int32_t tile_log2 = 0;
    while ((1 << tile_log2) < tile_cols) ++tile_log2;
    const int64_t n_tiles64 = B->n_rows == 0 ? 1 : ((B->n_rows + tile_cols - 1) >> tile_log2);
    const int64_t n_bins = B->n_cols * n_tiles64;
    if (n_bins + 1 >= (int64_t)1 << 31) {
        sg_set_error("segment table of %lld x %lld entries is too large; use more right-hand blocks",
                     (long long)B->n_cols, (long long)n_tiles64);
        return SG_ERR_OVERFLOW;
    }
    sg_csr *permuted = nullptr;
    uint32_t *orig_of = nullptr, *pos_of = nullptr;
    SgTimer *timer = (flags & SG_POSTINGS_INNER) ? nullptr : new (std::nothrow) SgTimer(ctx, SG_K_POSTINGS);   // the whole build, the permuted copy included
    struct TimerGuard {
        SgTimer *t;
        ~TimerGuard() { delete t; }
    } timer_guard{timer};
    // the packed rows of the pruned multiply are written by the same pass that copies the rows into position order
    void *early_fwd = nullptr;
    uint32_t *early_fwd_ptr = nullptr;
    void *early_q8 = nullptr;   // ... and so are the 8-bit copies of the second filter
    const float norm_up_build = __builtin_nextafterf(sqrtf(max_norm2) * 1.000001f, 2.f);   // (= p->norm_up below)
    // second filter: terms must fit 24 bits; a sixteenth of the device memory at most; SG_Q8=0 switches it off
    // ... and rows of a name list's length: the filter walks the candidate's entries like the exact scoring does and saves its
    // memory round trips -- on rows of 60 entries (a record of two lines, fifteen units) it costs more than it saves (100 k
    // long names, SG_Q8 = 1 / 0: 8.9 / 5.8 ms; profiles/r05_family_sweep_q8.log).  SG_Q8=1 forces it for any length.
    // Where the bar lies (round 6, scripts/q8_band_sweep.py, profiles/r06b_q8_band_sweep.log: 100 k rows cut to 16 .. 57 entries a
    // row): with the records is the faster up to 45 entries a row (by 3 - 30 %), level at 50, and 50 - 75 % slower at 57 (rows
    // beyond 60 entries have no copy and pass unseen, a record of two lines costs fifteen units) -- 40 in round 5, 45 now.
    const bool q8_forced = ctx->opt("SG_Q8") && ctx->opt("SG_Q8")[0] == '1';
    const bool want_q8 = want_pruned && !(flags & SG_POSTINGS_TILE_FORM) && B_in->n_cols < ((int64_t)1 << 24) && !(ctx->opt("SG_Q8") && ctx->opt("SG_Q8")[0] == '0') &&
                         (q8_forced || (double)B_in->nnz <= 45.0 * (double)B_in->n_rows) &&
                         (ctx->total_mem == 0 || (size_t)SG_Q8_STRIDE * ((size_t)B_in->n_rows + 1) < ctx->total_mem / 16);
    bool fwd_done = false;
    bool aux_written = false;   // the scoring context and the null postings were written by the tables kernel
    {
        const bool filt = want_pruned && sg_pruned_supports_tile(tile_log2) &&
                          (B->n_cols + 1) * ((n_tiles64 + 3) & ~(int64_t)3) < ((int64_t)1 << 30);
        const bool blk = ctx->opt("SG_ROW_BLOCKS") && ctx->opt("SG_ROW_BLOCKS")[0] == '1';
        if (filt && !blk && !(flags & SG_POSTINGS_NO_PERMUTATION) && B_in->n_rows > 0) {
            int st0 = ctx->alloc(((size_t)B_in->nnz + 8) * (B_in->dtype == SG_F64 ? 16 : 8), &early_fwd);
            if (st0 == SG_OK) st0 = sg_alloc(ctx, 2 * ((size_t)B_in->n_rows + 2), &early_fwd_ptr);
            if (st0 == SG_OK && want_q8) st0 = ctx->alloc((size_t)SG_Q8_STRIDE * ((size_t)B_in->n_rows + 1), &early_q8);
            if (st0 != SG_OK) {
                ctx->release(early_fwd);
                ctx->release(early_fwd_ptr);
                return st0;
            }
        }
    }
    if (!(flags & SG_POSTINGS_NO_PERMUTATION)) {
        const int stp = build_permuted(ctx, B_in, tile_cols, &permuted, &orig_of, &pos_of, pending, early_fwd_ptr, early_fwd, &fwd_done,
                                       early_q8, 1.0f / norm_up_build);
        if (stp != SG_OK) {
            ctx->release(early_fwd);
            ctx->release(early_fwd_ptr);
            ctx->release(early_q8);
            return stp;
        }
    }
    if (!fwd_done) {
        ctx->release(early_fwd);
        ctx->release(early_fwd_ptr);
        ctx->release(early_q8);
        early_fwd = nullptr;
        early_fwd_ptr = nullptr;
        early_q8 = nullptr;
    }
    if (pending && pending->pending_src && !permuted) {      // no copy in position order was made: the index reads the representatives' matrix itself
        const int stm = sg_collapse_materialize(ctx, pending);
        if (stm != SG_OK) {
            sg_csr_free(permuted);
            ctx->release(orig_of);
            ctx->release(pos_of);
            return stm;
        }
    }
    if (permuted) B = permuted;   // everything below indexes right-hand rows by POSITION
    sg_postings *p = new (std::nothrow) sg_postings();
    if (!p) {
        sg_csr_free(permuted);
        ctx->release(orig_of);
        ctx->release(pos_of);
        ctx->release(early_fwd);
        ctx->release(early_fwd_ptr);
        ctx->release(early_q8);
        return SG_ERR_OOM;
    }
    p->permuted = permuted;
    p->d_orig_of = orig_of;
    p->d_pos_of = pos_of;
    p->ctx = ctx;
    p->n_right = B->n_rows;
    p->n_terms = B->n_cols;
    p->nnz = B->nnz;
    p->dtype = B->dtype;
    p->tile_log2 = tile_log2;
    p->tile_form = (flags & SG_POSTINGS_TILE_FORM) != 0;
    p->built_from = *B_in;
    p->built_from.owned = false;
    p->built_from.d_props_words = nullptr;
    p->built_from.left_groups = nullptr;
    p->built_from.left_state = 0;
    p->built_from_valid = true;
    p->n_tiles = (int32_t)n_tiles64;
    p->b_indptr = B_in->d_indptr;      // the caller's matrix: what "A is the matrix the postings were built from" compares
    p->b_indices = B_in->d_indices;
    p->b_data = B_in->d_data;
    p->cosine_like = cosine_like;
    p->max_norm2 = max_norm2;
    const size_t vs = 8;   // f64 value, or packed {row, f32 value}
    int st = sg_alloc(ctx, (size_t)n_bins + 1, &p->d_seg);
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)B->n_cols + 1, &p->d_term_len);
    // The postings proper ({accumulator slot, value}: 8 bytes each, 100 MB at 663 k) are what the EXACT kernel streams.  When
    // the pruned multiply will take the product they are only needed for the rows it hands over (wider than 128 non-zeros
    // ...: none in a list of names), and scattering them is half of the index build: they are then written on demand
    // (sg_postings_ensure_full), from the same segment table.
    const bool will_filter = want_pruned && sg_pruned_supports_tile(tile_log2) &&
                             (B->n_cols + 1) * ((n_tiles64 + 3) & ~(int64_t)3) < ((int64_t)1 << 30);
    const size_t lds_need = (size_t)B->n_cols * 4 + ((size_t)(B->n_cols + 31) / 32) * 4;
    bool lds_path = B->n_rows > 0 && lds_need <= 124 * 1024 && B->n_cols > 0;
    if (const char *e = ctx->opt("SG_POSTINGS_LDS")) lds_path = lds_path && e[0] != '0';
    const bool lazy_full = will_filter && lds_path && !(ctx->opt("SG_POSTINGS_LAZY") && ctx->opt("SG_POSTINGS_LAZY")[0] == '0');
    p->src = *B;
    p->src.owned = false;
    p->src.d_props_words = nullptr;
    p->src.left_groups = nullptr;
    p->src.left_state = 0;
    if (st == SG_OK && !lazy_full && B->dtype == SG_F64) st = sg_alloc(ctx, (size_t)B->nnz + 64, &p->d_rows);
    if (st == SG_OK && !lazy_full) st = ctx->alloc(((size_t)B->nnz + 64) * vs, &p->d_vals);
    if (st == SG_OK && will_filter) {
        {
            // rows at a fixed stride for the exact scoring, when the longest row fits 1 KiB (127 entries f32 / 63 f64)
            uint32_t max_nnz = 0;
            bool cl = false;
            float n2 = 0.f;
            const bool want_blk = ctx->opt("SG_ROW_BLOCKS") && ctx->opt("SG_ROW_BLOCKS")[0] == '1';
            if (want_blk) {
                // the longest row is measured only for this (a vectoriser-made matrix, and what is derived from it, is known
                // to be cosine-like without a look: sg_csr_props)
                if (B->props_max_nnz == 0 && !B->d_props_words) B->props_state = 0;
                st = sg_csr_props(ctx, B, &cl, &n2, &max_nnz);
            }
            const size_t es = B->dtype == SG_F64 ? 16 : 8;
            const size_t need = (((size_t)max_nnz + 1) * es + 127) / 128 * 128;
            // Opt-in (SG_ROW_BLOCKS=1): the blocks cut the memory-side traffic of the multiply by a fifth (57.9 -> 43 GB per
            // launch at 663 k) but not its time -- the kernel is not bound by bytes -- and their scorer's extra loop
            // trips cost 0.3 - 0.7 ms (9.76 ms packed / 10.08 ms with 64-byte units / 10.50 ms with 32-byte units:
            // profiles/r03_row_blocks_ab.log); the index build pays 0.11 ms for them.
            if (st == SG_OK && want_blk && need <= 1024 && (double)need * (double)B->n_rows < 3.5e9 &&
                (ctx->total_mem == 0 || need * (size_t)B->n_rows < ctx->total_mem / 8)) {
                p->blk_bytes = (uint32_t)need;
                st = ctx->alloc(need * ((size_t)B->n_rows + 1), &p->d_blk);
            }
        }
        if (early_fwd && !p->d_blk) {          // written along with the rows' copy in position order (build_permuted)
            p->d_fwd = early_fwd;
            p->d_fwd_ptr = early_fwd_ptr;
            p->d_q8 = early_q8;
            early_fwd = nullptr;
            early_fwd_ptr = nullptr;
            early_q8 = nullptr;
        } else {
            if (st == SG_OK && !p->d_blk) st = ctx->alloc(((size_t)B->nnz + 8) * (B->dtype == SG_F64 ? 16 : 8), &p->d_fwd);
            if (st == SG_OK) st = sg_alloc(ctx, 2 * ((size_t)B->n_rows + 2), &p->d_fwd_ptr);   // uint2 per row
            if (st == SG_OK && want_q8 && !p->d_blk) st = ctx->alloc((size_t)SG_Q8_STRIDE * ((size_t)B->n_rows + 1), &p->d_q8);
            fwd_done = false;
        }
        // slack: the pruned multiply loads a lane's four slots of a segment unconditionally (<= 4 * 63 entries past it)
        if (st == SG_OK) st = sg_alloc(ctx, (size_t)B->nnz + 512, &p->d_filt);
        // the stream form points lanes without a posting at the slack behind the array: entries that add 0 (bq = 0), each to
        // an accumulator of its own (64 lanes adding to ONE LDS word are serialised: 3 ms at 663 k)
        // (written by postings_tables_kernel on the LDS build path, by a launch of its own otherwise: below)
        p->nt_pad = (int32_t)((n_tiles64 + 3) & ~(int64_t)3);
        if (st == SG_OK) st = sg_alloc(ctx, (size_t)(B->n_cols + 1) * (size_t)p->nt_pad + 4, &p->d_ends);
        // stream form of the pruned multiply (sg_spgemm_pruned.hip): eight tiles share one accumulator tile
        p->fold_log2 = 0;
        if (tile_log2 == 12 && !(ctx->opt("SG_K4_STREAM") && ctx->opt("SG_K4_STREAM")[0] == '0')) p->fold_log2 = 3;
        if (p->fold_log2 > 0) {
            const int64_t n_super = (n_tiles64 + ((int64_t)1 << p->fold_log2) - 1) >> p->fold_log2;
            p->nv_pad = (int32_t)((n_super + 3) & ~(int64_t)3);
            if (st == SG_OK) st = sg_alloc(ctx, (size_t)(B->n_cols + 1) * (size_t)p->nv_pad + 4, &p->d_ends8);
        }
        p->norm_up = norm_up_build;
        // a term is "frequent" when it occurs in at least this share of the right-hand rows: the suffix of a
        // left row is drawn from frequent terms only, which lets the survivor test use each candidate's own
        // frequent-part norm instead of 1 (profiles/r01_prune_tuning.log)
        double frac = 0.005;   // (round 5, with the second filter: 0.0045 -> 0.005, 30.8 M -> 22.5 M candidates, kernel - 2 % at 663 k; 0.004 + 5 %, 0.007 + 4 %)
        if (const char *v = ctx->opt("SG_PRUNE_FREQ")) frac = atof(v);
        const double fm = frac * (double)B->n_rows;
        p->freq_min = fm < 1.0 ? 1u : (uint32_t)fm;
    }
    ctx->release(early_fwd);      // (only when the build took another turn than the one they were made for)
    ctx->release(early_fwd_ptr);
    ctx->release(early_q8);
    if (st != SG_OK) {
        sg_postings_free(p);
        return st;
    }
    {
        // one thread per slot of the (tile x row-in-tile) grid: covers every row, see row_of_thread
        const unsigned grid = (unsigned)((((int64_t)p->n_tiles << tile_log2) + 255) / 256);
        // the tile's counters fit in LDS: one workgroup per tile (part), LDS atomics (otherwise global ones)
        const size_t lds = (size_t)B->n_cols * 4 + ((size_t)(B->n_cols + 31) / 32) * 4;
        bool in_lds = B->n_rows > 0 && lds <= 124 * 1024 && B->n_cols > 0;
        if (const char *e = ctx->opt("SG_POSTINGS_LDS")) in_lds = in_lds && e[0] != '0';
        const float inv_norm = p->d_filt ? 1.0f / p->norm_up : 0.f;
        bool tables_done = false;
        if (in_lds) {
            static bool attr_done = false;
            if (!attr_done) {
                (void)hipFuncSetAttribute((const void *)postings_count_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 124 * 1024);
                (void)hipFuncSetAttribute((const void *)postings_fill_lds<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 124 * 1024);
                (void)hipFuncSetAttribute((const void *)postings_fill_lds<double>, hipFuncAttributeMaxDynamicSharedMemorySize, 124 * 1024);
                (void)hipFuncSetAttribute((const void *)postings_fill_staged<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                (void)hipFuncSetAttribute((const void *)postings_fill_staged<double>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                attr_done = true;
            }
            // Round 6: filter postings staged in LDS and written cell by cell (postings_fill_staged) -- when only the filter
            // postings are written (the exact kernel's are lazy) and a chunk of at least 64 rows fits the stage.  The stage
            // takes what the packed counters leave of 158 KiB: one workgroup per CU, so tiles are split until there are two
            // workgroups per CU (parts of >= 512 rows) and a launch does not end with a few CUs working alone.
            bool staged = p->d_filt != nullptr && p->d_vals == nullptr && !(ctx->opt("SG_FILL_STAGED") && ctx->opt("SG_FILL_STAGED")[0] == '0');
            uint32_t stage_cap = 0;
            int32_t chunk_rows = 0;
            size_t staged_lds = 0;
            const double mean_nnz = B->n_rows > 0 ? (double)B->nnz / (double)B->n_rows : 1.0;
            if (staged) {
                const size_t fixed0 = ((((size_t)B->n_cols + 1) / 2 + 3) & ~(size_t)3) * 4 + (((size_t)B->n_cols + 31) / 32) * 4 + 32 * 4;
                chunk_rows = 2048;
                for (;;) {
                    const size_t fixed = fixed0 + (size_t)chunk_rows * 4;
                    const size_t room = fixed + 16384 < 158 * 1024 ? 158 * 1024 - fixed : 0;
                    stage_cap = (uint32_t)(room / 4);
                    if (stage_cap > 65535u) stage_cap = 65535u;
                    if ((double)chunk_rows * mean_nnz * 1.2 + 64.0 <= (double)stage_cap || chunk_rows <= 64) break;
                    chunk_rows >>= 1;
                }
                staged = (double)chunk_rows * mean_nnz * 1.2 + 64.0 <= (double)stage_cap;
                staged_lds = fixed0 + (size_t)chunk_rows * 4 + (size_t)stage_cap * 4;
                if (const char *v = ctx->opt("SG_FILL_STAGE_CAP"))    // test hook: chunks that do not fit go posting by posting
                    if (atoi(v) > 0 && (uint32_t)atoi(v) < stage_cap) stage_cap = (uint32_t)atoi(v);
            }
            // fewer tiles than CUs: split every tile between 2 or 4 workgroups (parts of >= 1024 rows)
            int32_t split = 1;
            while (split < 4 && (int64_t)p->n_tiles * split < ctx->num_cu && (tile_cols / (split * 2)) >= 1024 &&
                   (n_bins * split * 2 + 1) < ((int64_t)1 << 31))
                split *= 2;
            if (staged)
                while (split < 8 && (int64_t)p->n_tiles * split < 2 * (int64_t)ctx->num_cu && (tile_cols / (split * 2)) >= 512 &&
                       (n_bins * split * 2 + 1) < ((int64_t)1 << 31))
                    split *= 2;
            if (const char *e = ctx->opt("SG_POSTINGS_SPLIT")) {
                const int o = atoi(e);
                if ((o == 1 || o == 2 || o == 4) && tile_cols / o >= 64 && n_bins * o + 1 < ((int64_t)1 << 31)) split = o;
            }
            p->split = split;
            const int64_t wgs64 = (int64_t)p->n_tiles * split;
            uint32_t *cnt = nullptr;            // [workgroup][term] (see postings_count_lds)
            uint8_t *is_frequent = nullptr;
            st = sg_alloc(ctx, (size_t)(wgs64 * B->n_cols) + 1, &cnt);
            if (st == SG_OK) st = sg_alloc(ctx, (size_t)B->n_cols + 4, &is_frequent);
            if (st == SG_OK) st = sg_alloc(ctx, (size_t)B->n_cols + 2, &p->d_term_start);
            if (st == SG_OK) {
                const unsigned wgs = (unsigned)wgs64;
                const unsigned strips = (unsigned)((B->n_cols + 63) / 64), strips1 = (unsigned)((B->n_cols + 64) / 64);
                hipLaunchKernelGGL(postings_count_lds, dim3(wgs), dim3(1024), (size_t)B->n_cols * 4, ctx->stream, B->d_indptr,
                                   B->d_indices, B->n_rows, tile_log2, (int32_t)B->n_cols, split, cnt);
                hipLaunchKernelGGL(postings_colscan_kernel, dim3(strips), dim3(1024), 0, ctx->stream, cnt, (int32_t)wgs64,
                                   (int32_t)B->n_cols, p->freq_min, p->d_term_len, is_frequent);
                // the terms' lists back to back: starts of the lists, the last entry receives the total (= nnz)
                st = sg_exclusive_scan_u32(ctx, p->d_term_len, p->d_term_start, B->n_cols, p->d_term_start + B->n_cols);
                if (st == SG_OK) {
                    if (staged)
                        ;     // (after the tables: the staged fill advances the workgroups' rows of `cnt`)
                    else if (B->dtype == SG_F64)
                        hipLaunchKernelGGL(postings_fill_lds<double>, dim3(wgs), dim3(1024), lds, ctx->stream, B->d_indptr,
                                           B->d_indices, (const double *)B->d_data, B->n_rows, tile_log2, (int32_t)B->n_cols, split,
                                           (const uint32_t *)cnt, (const uint32_t *)p->d_term_start, (const uint8_t *)is_frequent,
                                           p->d_rows, (double *)p->d_vals, p->d_filt, inv_norm, p->fold_log2);
                    else
                        hipLaunchKernelGGL(postings_fill_lds<float>, dim3(wgs), dim3(1024), lds, ctx->stream, B->d_indptr,
                                           B->d_indices, (const float *)B->d_data, B->n_rows, tile_log2, (int32_t)B->n_cols, split,
                                           (const uint32_t *)cnt, (const uint32_t *)p->d_term_start, (const uint8_t *)is_frequent,
                                           p->d_rows, (float *)p->d_vals, p->d_filt, inv_norm, p->fold_log2);
                    SgScoreCtx sc;
                    if (p->d_fwd_ptr) {
                        st = ctx->alloc(256, (void **)&p->d_score_ctx);
                        sc.fwd_ptr = p->d_fwd_ptr;
                        sc.fwd = p->d_fwd;
                        sc.blk = p->d_blk;
                        sc.blk_bytes = p->blk_bytes;
                        sc.orig_of = p->d_orig_of;
                        sc.q8 = (const uint4 *)p->d_q8;
                        sc.q8_scale = p->d_q8 ? __builtin_nextafterf((float)(255.0 / (double)p->norm_up * (1.0 - 1e-6)), 0.f) : 0.f;
                    }
                    if (st == SG_OK) {
                        hipLaunchKernelGGL(postings_tables_kernel, dim3(strips1), dim3(1024), 0, ctx->stream, (const uint32_t *)cnt,
                                           (const uint32_t *)p->d_term_start, (int32_t)B->n_cols, p->n_tiles, split, p->d_seg, p->d_ends,
                                           p->nt_pad, p->d_ends8, p->nv_pad, p->fold_log2, sc, p->d_score_ctx,
                                           p->d_filt ? p->d_filt + B->nnz : (uint32_t *)nullptr);
                        if (staged && B->dtype == SG_F64)
                            hipLaunchKernelGGL(postings_fill_staged<double>, dim3(wgs), dim3(1024), staged_lds, ctx->stream, B->d_indptr,
                                               B->d_indices, (const double *)B->d_data, B->n_rows, tile_log2, (int32_t)B->n_cols, split,
                                               chunk_rows, stage_cap, cnt, (const uint32_t *)p->d_term_start, (const uint8_t *)is_frequent,
                                               p->d_filt, inv_norm, p->fold_log2);
                        else if (staged)
                            hipLaunchKernelGGL(postings_fill_staged<float>, dim3(wgs), dim3(1024), staged_lds, ctx->stream, B->d_indptr,
                                               B->d_indices, (const float *)B->d_data, B->n_rows, tile_log2, (int32_t)B->n_cols, split,
                                               chunk_rows, stage_cap, cnt, (const uint32_t *)p->d_term_start, (const uint8_t *)is_frequent,
                                               p->d_filt, inv_norm, p->fold_log2);
                        if (hipGetLastError() != hipSuccess) st = SG_ERR_HIP;
                        aux_written = st == SG_OK;
                    }
                }
            }
            ctx->release(cnt);
            ctx->release(is_frequent);
            tables_done = true;
        } else {
            uint32_t *cursor = nullptr;
            st = sg_alloc(ctx, (size_t)n_bins + 1, &cursor);
            if (st == SG_OK) {
                SG_HIP_TRY(hipMemsetAsync(p->d_seg, 0, sizeof(uint32_t) * (size_t)(n_bins + 1), ctx->stream));
                SG_HIP_TRY(hipMemsetAsync(cursor, 0, sizeof(uint32_t) * (size_t)(n_bins + 1), ctx->stream));
            }
            if (st == SG_OK && grid > 0) {
                if (B->dtype == SG_F64)
                    hipLaunchKernelGGL(postings_count<double>, dim3(grid), dim3(256), 0, ctx->stream, B->d_indptr,
                                       B->d_indices, B->n_rows, tile_log2, p->n_tiles, p->d_seg);
                else
                    hipLaunchKernelGGL(postings_count<float>, dim3(grid), dim3(256), 0, ctx->stream, B->d_indptr,
                                       B->d_indices, B->n_rows, tile_log2, p->n_tiles, p->d_seg);
                SG_HIP_TRY(hipGetLastError());
            }
            // counts -> offsets, in place; seg[n_bins] receives the total (= nnz)
            if (st == SG_OK) st = sg_exclusive_scan_u32(ctx, p->d_seg, p->d_seg, n_bins, p->d_seg + n_bins);
            if (st == SG_OK && grid > 0) {
                if (B->dtype == SG_F64)
                    hipLaunchKernelGGL(postings_fill<double>, dim3(grid), dim3(256), 0, ctx->stream, B->d_indptr,
                                       B->d_indices, (const double *)B->d_data, B->n_rows, tile_log2, p->n_tiles,
                                       p->d_seg, cursor, p->d_rows, (double *)p->d_vals, p->d_filt, p->freq_min, inv_norm, p->fold_log2);
                else
                    hipLaunchKernelGGL(postings_fill<float>, dim3(grid), dim3(256), 0, ctx->stream, B->d_indptr,
                                       B->d_indices, (const float *)B->d_data, B->n_rows, tile_log2, p->n_tiles,
                                       p->d_seg, cursor, p->d_rows, (float *)p->d_vals, p->d_filt, p->freq_min, inv_norm, p->fold_log2);
                if (hipGetLastError() != hipSuccess) st = SG_ERR_HIP;
            }
            ctx->release(cursor);
        }
        if (st == SG_OK && !tables_done && B->n_cols > 0) {
            // (the path with global counters builds the term-major table itself; lists and ends are read off it)
            hipLaunchKernelGGL(term_len_kernel, dim3((unsigned)((B->n_cols + 255) / 256)), dim3(256), 0, ctx->stream,
                               (const uint32_t *)p->d_seg, B->n_cols, p->n_tiles, p->d_term_len);
            if (p->d_ends || p->d_ends8)
                hipLaunchKernelGGL(ends_from_seg_kernel, dim3((unsigned)((B->n_cols + 256) / 256)), dim3(256), 0, ctx->stream,
                                   (const uint32_t *)p->d_seg, B->n_cols, p->n_tiles, p->d_ends, p->nt_pad, p->d_ends8, p->nv_pad,
                                   p->fold_log2);
            if (hipGetLastError() != hipSuccess) st = SG_ERR_HIP;
        }
        if (st == SG_OK && p->d_fwd && !fwd_done) {   // (packed rows in use: no row blocks; the fused pass writes the pad itself)
            // the exact scoring reads packed rows in rounds of eight entries and multiplies the slots past a row's end by
            // a = 0: the pad behind the LAST row must hold finite values (0 * NaN would poison that row's score)
            const size_t es = B->dtype == SG_F64 ? 16 : 8;
            if (hipMemsetAsync((char *)p->d_fwd + (size_t)B->nnz * es, 0, 8 * es, ctx->stream) != hipSuccess) st = SG_ERR_HIP;
        }
        if (st == SG_OK && p->d_blk && B->n_rows > 0) {
            const int64_t work = B->n_rows * (int64_t)(p->blk_bytes / 16);
            const unsigned g3 = (unsigned)((work + 255) / 256);
            if (B->dtype == SG_F64)
                hipLaunchKernelGGL(row_blocks_kernel<double>, dim3(g3), dim3(256), 0, ctx->stream, B->d_indptr, B->d_indices,
                                   (const double *)B->d_data, B->n_rows, (const uint32_t *)p->d_orig_of, p->blk_bytes, p->d_blk);
            else
                hipLaunchKernelGGL(row_blocks_kernel<float>, dim3(g3), dim3(256), 0, ctx->stream, B->d_indptr, B->d_indices,
                                   (const float *)B->d_data, B->n_rows, (const uint32_t *)p->d_orig_of, p->blk_bytes, p->d_blk);
            if (hipGetLastError() != hipSuccess) st = SG_ERR_HIP;
        }
        if (st == SG_OK && p->d_fwd_ptr && !fwd_done) {
            const int64_t work = B->nnz > B->n_rows + 1 ? B->nnz : B->n_rows + 1;
            const unsigned g2 = (unsigned)((work + 255) / 256);
            if (B->dtype == SG_F64)
                hipLaunchKernelGGL(fwd_pack<double>, dim3(g2), dim3(256), 0, ctx->stream, B->d_indptr, B->d_indices,
                                   (const double *)B->d_data, B->n_rows, B->nnz, (const uint32_t *)p->d_orig_of, p->d_fwd_ptr, p->d_fwd);
            else
                hipLaunchKernelGGL(fwd_pack<float>, dim3(g2), dim3(256), 0, ctx->stream, B->d_indptr, B->d_indices,
                                   (const float *)B->d_data, B->n_rows, B->nnz, (const uint32_t *)p->d_orig_of, p->d_fwd_ptr, p->d_fwd);
            if (p->d_q8 && B->n_rows > 0) {
                const unsigned g4 = (unsigned)((B->n_rows * 16 + 255) / 256);
                if (B->dtype == SG_F64)
                    hipLaunchKernelGGL(q8_pack_kernel<double>, dim3(g4), dim3(256), 0, ctx->stream, B->d_indptr, B->d_indices,
                                       (const double *)B->d_data, B->n_rows, (const uint32_t *)p->d_orig_of, (uint4 *)p->d_q8, 1.0f / p->norm_up);
                else
                    hipLaunchKernelGGL(q8_pack_kernel<float>, dim3(g4), dim3(256), 0, ctx->stream, B->d_indptr, B->d_indices,
                                       (const float *)B->d_data, B->n_rows, (const uint32_t *)p->d_orig_of, (uint4 *)p->d_q8, 1.0f / p->norm_up);
            }
            if (hipGetLastError() != hipSuccess) st = SG_ERR_HIP;
        }
    }
    if (st == SG_OK && p->d_filt && !aux_written) {
        hipLaunchKernelGGL(null_postings_kernel, dim3(2), dim3(256), 0, ctx->stream, p->d_filt + B->nnz);
        if (hipGetLastError() != hipSuccess) st = SG_ERR_HIP;
    }
    if (st == SG_OK && p->d_fwd_ptr && !aux_written) {
        st = ctx->alloc(256, (void **)&p->d_score_ctx);
        if (st == SG_OK) {
            SgScoreCtx sc;
            sc.fwd_ptr = p->d_fwd_ptr;
            sc.fwd = p->d_fwd;
            sc.blk = p->d_blk;
            sc.blk_bytes = p->blk_bytes;
            sc.orig_of = p->d_orig_of;
            sc.q8 = (const uint4 *)p->d_q8;
            sc.q8_scale = p->d_q8 ? __builtin_nextafterf((float)(255.0 / (double)p->norm_up * (1.0 - 1e-6)), 0.f) : 0.f;
            hipLaunchKernelGGL(score_ctx_kernel, dim3(1), dim3(1), 0, ctx->stream, sc, p->d_score_ctx);
            if (hipGetLastError() != hipSuccess) st = SG_ERR_HIP;
        }
    }
    if (st != SG_OK) {
        sg_postings_free(p);
        return st;
    }
    *out = p;
    return SG_OK;
}

// The postings proper, when the build left them out (see sg_postings_build_flags): same two passes, values only.
int sg_postings_ensure_full(sg_ctx *ctx, const sg_postings *cp) {
    sg_postings *p = const_cast<sg_postings *>(cp);
    if (p->d_vals || p->nnz <= 0) return SG_OK;
    const sg_csr *B = &p->src;
    const int64_t n_bins = p->n_terms * (int64_t)p->n_tiles;
    const int32_t split = p->split > 0 ? p->split : 1;
    int st = SG_OK;
    if (B->dtype == SG_F64) st = sg_alloc(ctx, (size_t)B->nnz + 64, &p->d_rows);
    if (st == SG_OK) st = ctx->alloc(((size_t)B->nnz + 64) * 8, &p->d_vals);
    uint32_t *cnt = nullptr, *len_scratch = nullptr;
    const int64_t wgs64 = (int64_t)p->n_tiles * split;
    (void)n_bins;
    if (st == SG_OK && !p->d_term_start) {
        sg_set_error("postings were built without the LDS path: nothing is lazy there");
        st = SG_ERR_BADARG;
    }
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)(wgs64 * B->n_cols) + 1, &cnt);
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)B->n_cols + 2, &len_scratch);
    if (st == SG_OK) {
        const unsigned wgs = (unsigned)wgs64;
        const size_t lds = (size_t)B->n_cols * 4 + ((size_t)(B->n_cols + 31) / 32) * 4;
        hipLaunchKernelGGL(postings_count_lds, dim3(wgs), dim3(1024), (size_t)B->n_cols * 4, ctx->stream, B->d_indptr, B->d_indices,
                           B->n_rows, p->tile_log2, (int32_t)B->n_cols, split, cnt);
        hipLaunchKernelGGL(postings_colscan_kernel, dim3((unsigned)((B->n_cols + 63) / 64)), dim3(1024), 0, ctx->stream, cnt,
                           (int32_t)wgs64, (int32_t)B->n_cols, 0u, len_scratch, (uint8_t *)nullptr);
        if (B->dtype == SG_F64)
            hipLaunchKernelGGL(postings_fill_lds<double>, dim3(wgs), dim3(1024), lds, ctx->stream, B->d_indptr, B->d_indices,
                               (const double *)B->d_data, B->n_rows, p->tile_log2, (int32_t)B->n_cols, split, (const uint32_t *)cnt,
                               (const uint32_t *)p->d_term_start, (const uint8_t *)nullptr, p->d_rows, (double *)p->d_vals,
                               (uint32_t *)nullptr, 0.f, 0);
        else
            hipLaunchKernelGGL(postings_fill_lds<float>, dim3(wgs), dim3(1024), lds, ctx->stream, B->d_indptr, B->d_indices,
                               (const float *)B->d_data, B->n_rows, p->tile_log2, (int32_t)B->n_cols, split, (const uint32_t *)cnt,
                               (const uint32_t *)p->d_term_start, (const uint8_t *)nullptr, p->d_rows, (float *)p->d_vals,
                               (uint32_t *)nullptr, 0.f, 0);
        if (hipGetLastError() != hipSuccess) st = SG_ERR_HIP;
    }
    ctx->release(cnt);
    ctx->release(len_scratch);
    if (st != SG_OK) {
        ctx->release(p->d_vals);
        ctx->release(p->d_rows);
        p->d_vals = nullptr;
        p->d_rows = nullptr;
    }
    return st;
}

extern "C" int sg_postings_free(sg_postings *p) {
    if (!p) return SG_OK;
    p->ctx->release(p->d_seg);
    p->ctx->release(p->d_term_len);
    p->ctx->release(p->d_term_start);
    p->ctx->release(p->d_rows);
    p->ctx->release(p->d_vals);
    p->ctx->release(p->d_fwd);
    p->ctx->release(p->d_blk);
    p->ctx->release(p->d_q8);
    p->ctx->release(p->d_fwd_ptr);
    p->ctx->release(p->d_filt);
    p->ctx->release(p->d_ends);
    p->ctx->release(p->d_ends8);
    p->ctx->release(p->d_score_ctx);
    p->ctx->release(p->d_orig_of);
    p->ctx->release(p->d_pos_of);
    sg_csr_free(p->permuted);
    sg_collapse_free(p->collapse);
    if (p->plain) sg_postings_free(p->plain);
    if (p->exact_native) sg_postings_free(p->exact_native);
    if (p->alt_tile) sg_postings_free(p->alt_tile);
    delete p;
    return SG_OK;
}
