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

// norm of a row's frequent part (terms whose list holds >= freq_min entries), quantised upwards to 8 bits
// relative to norm_up
template <typename T>
__device__ __forceinline__ uint32_t frequent_norm_q8(const int32_t *__restrict__ indices, const T *__restrict__ data,
                                                     int64_t lo, int64_t hi, const uint32_t *__restrict__ seg, int32_t n_tiles,
                                                     uint32_t freq_min, float inv_norm_up) {
    double f2 = 0.0;
    for (int64_t p = lo; p < hi; ++p) {
        const int64_t k = indices[p];
        if (seg[(k + 1) * n_tiles] - seg[k * n_tiles] >= freq_min) f2 += (double)data[p] * (double)data[p];
    }
    uint32_t fq = (uint32_t)ceilf(__double2float_ru(sqrt(f2)) * inv_norm_up * 255.0f * 1.000002f);
    return fq > 255u ? 255u : fq;
}

// one posting (+ its filter posting) of column `col` of the tile at position `pos`.
// Filter posting (read by K4p, sg_spgemm_pruned.hip), 32 bits, AB = tile_log2 + 1:
//   [0]        h     which 16-bit half of the accumulator word the column owns (col & 1)
//   [1]        0
//   [2, AB)    word  (col >> 1): bits [0, AB) masked with ~3 ARE the byte address of the accumulator word in LDS
//   [AB, 24)   bq    value quantised upwards relative to norm_up
//   [24, 32)   fq    norm of the row's frequent part, quantised upwards relative to norm_up
template <typename T>
__device__ __forceinline__ void emit_posting(int32_t *out_rows, T *out_vals, uint32_t *out_filt, uint32_t pos, uint32_t col,
                                             T v, uint32_t fq, int32_t tile_log2, float inv_norm_up) {
    // the multiply wants the byte offset of the accumulator inside its LDS tile, not j itself
    store_posting<T>(out_rows, out_vals, pos, (int32_t)(col * (uint32_t)sizeof(T)), v);
    if (out_filt) {
        const int32_t ab = tile_log2 + 1;
        const uint32_t bq_max = (1u << (24 - ab)) - 1u;   // the bits the address and fq leave
        uint32_t bq = (uint32_t)ceilf((float)v * inv_norm_up * (float)bq_max * 1.000002f);
        if (bq > bq_max) bq = bq_max;
        out_filt[pos] = ((col >> 1) << 2) | (col & 1u) | (bq << ab) | (fq << 24);
    }
}

template <typename T>
__global__ void __launch_bounds__(256) postings_fill(const int64_t *__restrict__ indptr,
                                                     const int32_t *__restrict__ indices,
                                                     const T *__restrict__ data, int64_t n_rows, int32_t tile_log2,
                                                     int32_t n_tiles, const uint32_t *__restrict__ seg,
                                                     uint32_t *cursor, int32_t *out_rows, T *out_vals,
                                                     uint32_t *out_filt /* null: no filter postings */,
                                                     uint32_t freq_min, float inv_norm_up) {
    const int64_t j = row_of_thread((int64_t)blockIdx.x * blockDim.x + threadIdx.x, tile_log2, n_tiles);
    if (j >= n_rows) return;
    const int64_t lo = indptr[j], hi = indptr[j + 1];
    const uint32_t t = (uint32_t)(j >> tile_log2);
    const uint32_t col = (uint32_t)(j & (((int64_t)1 << tile_log2) - 1));
    const uint32_t fq = out_filt ? frequent_norm_q8<T>(indices, data, lo, hi, seg, n_tiles, freq_min, inv_norm_up) : 0u;
    for (int64_t p = lo; p < hi; ++p) {
        const int64_t bin = (int64_t)indices[p] * n_tiles + t;
        const uint32_t pos = seg[bin] + atomicAdd(&cursor[bin], 1u);
        emit_posting<T>(out_rows, out_vals, out_filt, pos, col, data[p], fq, tile_log2, inv_norm_up);
    }
}

// The same two passes with the tile's counters in LDS: one workgroup per column tile (its rows are
// consecutive), a histogram / cursor array over the vocabulary in LDS (4 bytes per term), LDS atomics
// instead of 2 x nnz global ones.  Used when the vocabulary fits (n_terms * 4 <= 120 KiB).
template <typename T>
__global__ void __launch_bounds__(1024) postings_count_lds(const int64_t *__restrict__ indptr,
                                                           const int32_t *__restrict__ indices, int64_t n_rows,
                                                           int32_t tile_log2, int32_t n_tiles, int32_t n_terms,
                                                           uint32_t *seg_counts) {
    extern __shared__ uint32_t hist[];
    const int64_t t = blockIdx.x;
    for (int k = threadIdx.x; k < n_terms; k += blockDim.x) hist[k] = 0;
    __syncthreads();
    const int64_t j0 = t << tile_log2;
    int64_t j1 = j0 + ((int64_t)1 << tile_log2);
    if (j1 > n_rows) j1 = n_rows;
    for (int64_t j = j0 + threadIdx.x; j < j1; j += blockDim.x)
        for (int64_t p = indptr[j]; p < indptr[j + 1]; ++p) atomicAdd(&hist[indices[p]], 1u);
    __syncthreads();
    for (int k = threadIdx.x; k < n_terms; k += blockDim.x) seg_counts[(int64_t)k * n_tiles + t] = hist[k];
}

template <typename T>
__global__ void __launch_bounds__(1024) postings_fill_lds(const int64_t *__restrict__ indptr,
                                                          const int32_t *__restrict__ indices,
                                                          const T *__restrict__ data, int64_t n_rows, int32_t tile_log2,
                                                          int32_t n_tiles, int32_t n_terms, const uint32_t *__restrict__ seg,
                                                          int32_t *out_rows, T *out_vals, uint32_t *out_filt,
                                                          uint32_t freq_min, float inv_norm_up) {
    extern __shared__ uint32_t cursor[];   // next free slot of (term k, this tile)
    const int64_t t = blockIdx.x;
    for (int k = threadIdx.x; k < n_terms; k += blockDim.x) cursor[k] = seg[(int64_t)k * n_tiles + t];
    __syncthreads();
    const int64_t j0 = t << tile_log2;
    int64_t j1 = j0 + ((int64_t)1 << tile_log2);
    if (j1 > n_rows) j1 = n_rows;
    for (int64_t j = j0 + threadIdx.x; j < j1; j += blockDim.x) {
        const int64_t lo = indptr[j], hi = indptr[j + 1];
        const uint32_t col = (uint32_t)(j - j0);
        const uint32_t fq = out_filt ? frequent_norm_q8<T>(indices, data, lo, hi, seg, n_tiles, freq_min, inv_norm_up) : 0u;
        for (int64_t p = lo; p < hi; ++p) {
            const uint32_t pos = atomicAdd(&cursor[indices[p]], 1u);
            emit_posting<T>(out_rows, out_vals, out_filt, pos, col, data[p], fq, tile_log2, inv_norm_up);
        }
    }
}

// Packed copy of B's rows for the pruned multiply (one 16-byte load = two f32 entries or one f64 entry).
template <typename T>
__global__ void __launch_bounds__(256) fwd_pack(const int64_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                                                const T *__restrict__ data, int64_t n_rows, uint32_t *__restrict__ fwd_ptr,
                                                void *__restrict__ fwd) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j > n_rows) return;
    const int64_t base = indptr[0];
    fwd_ptr[j] = (uint32_t)(indptr[j] - base);
    if (j == n_rows) return;
    for (int64_t p = indptr[j]; p < indptr[j + 1]; ++p) {
        if (sizeof(T) == 4) {
            reinterpret_cast<int2 *>(fwd)[p - base] = make_int2(indices[p], __float_as_int((float)data[p]));
        } else {
            const long long bits = __double_as_longlong((double)data[p]);
            reinterpret_cast<int4 *>(fwd)[p - base] = make_int4(indices[p], 0, (int)(bits & 0xffffffffll), (int)(bits >> 32));
        }
    }
}

// segment ends of every term as byte offsets into the filter postings, rows padded to a multiple of four tiles
__global__ void __launch_bounds__(256) pack_ends_kernel(const uint32_t *__restrict__ seg, int64_t n_terms, int32_t n_tiles,
                                                        int32_t nt_pad, uint32_t *__restrict__ ends) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (n_terms + 1) * nt_pad) return;
    const int64_t k = i / nt_pad;
    if (k == n_terms) {   // one more row, all zero: the segments of a lane without a term (always empty)
        ends[i] = 0;
        return;
    }
    int32_t t = (int32_t)(i - k * nt_pad);
    if (t >= n_tiles) t = n_tiles - 1;
    ends[i] = seg[k * n_tiles + t + 1] << 2;
}

extern "C" int sg_postings_build(sg_ctx *ctx, const sg_csr *B, int32_t tile_cols, sg_postings **out) {
    SG_REQUIRE(ctx && B && out, "null argument");
    // cosine-like right-hand sides (non-negative, sorted rows, norms <= 1: TF-IDF) take the pruned multiply,
    // whose 16-bit accumulators make a 4096-column tile 8 KiB; everything else the exact kernel with 8 KiB
    // of float / double accumulators per wave
    bool cosine_like = false;
    float max_norm2 = 0.f;
    SG_TRY(sg_csr_props(ctx, B, &cosine_like, &max_norm2));
    const char *pr = getenv("SG_PRUNE");
    const bool want_pruned = cosine_like && !(pr && pr[0] == '0');
    if (tile_cols == 0) {
        tile_cols = B->dtype == SG_F64 ? 1024 : 2048;
        if (want_pruned) {
            tile_cols = 4096;
            if (const char *v = getenv("SG_PRUNE_TILE")) tile_cols = atoi(v) == 13 ? 8192 : (atoi(v) == 11 ? 2048 : 4096);
        }
    }
    SG_REQUIRE(tile_cols >= 256 && tile_cols <= 32768 && (tile_cols & (tile_cols - 1)) == 0,
               "tile_cols must be a power of two in [256, 32768]");
    int64_t max_entries = (int64_t)1 << 29;   // the multiply addresses postings with 32-bit BYTE offsets (8 B entries)
    if (const char *v = getenv("SG_MAX_POSTINGS")) {   // test hook: force the right-hand split at small sizes
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
    sg_postings *p = new (std::nothrow) sg_postings();
    if (!p) return SG_ERR_OOM;
    p->ctx = ctx;
    p->n_right = B->n_rows;
    p->n_terms = B->n_cols;
    p->nnz = B->nnz;
    p->dtype = B->dtype;
    p->tile_log2 = tile_log2;
    p->n_tiles = (int32_t)n_tiles64;
    p->b_indptr = B->d_indptr;
    p->b_indices = B->d_indices;
    p->b_data = B->d_data;
    p->cosine_like = cosine_like;
    p->max_norm2 = max_norm2;
    const size_t vs = 8;   // f64 value, or packed {row, f32 value}
    uint32_t *cursor = nullptr;
    int st = sg_alloc(ctx, (size_t)n_bins + 1, &p->d_seg);
    if (st == SG_OK && B->dtype == SG_F64) st = sg_alloc(ctx, (size_t)B->nnz + 64, &p->d_rows);
    if (st == SG_OK) st = ctx->alloc(((size_t)B->nnz + 64) * vs, &p->d_vals);
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)n_bins + 1, &cursor);
    if (st == SG_OK && want_pruned && sg_pruned_supports_tile(tile_log2) &&
        (B->n_cols + 1) * ((n_tiles64 + 3) & ~(int64_t)3) < ((int64_t)1 << 30)) {
        st = ctx->alloc(((size_t)B->nnz + 8) * (B->dtype == SG_F64 ? 16 : 8), &p->d_fwd);
        if (st == SG_OK) st = sg_alloc(ctx, (size_t)B->n_rows + 2, &p->d_fwd_ptr);
        // slack: the pruned multiply loads a lane's four slots of a segment unconditionally (<= 4 * 63 entries past it)
        if (st == SG_OK) st = sg_alloc(ctx, (size_t)B->nnz + 512, &p->d_filt);
        p->nt_pad = (int32_t)((n_tiles64 + 3) & ~(int64_t)3);
        if (st == SG_OK) st = sg_alloc(ctx, (size_t)(B->n_cols + 1) * (size_t)p->nt_pad + 4, &p->d_ends);
        p->norm_up = __builtin_nextafterf(sqrtf(max_norm2) * 1.000001f, 2.f);
        // a term is "frequent" when it occurs in at least this share of the right-hand rows: the suffix of a
        // left row is drawn from frequent terms only, which lets the survivor test use each candidate's own
        // frequent-part norm instead of 1 (profiles/r01_prune_tuning.log)
        double frac = 0.0045;
        if (const char *v = getenv("SG_PRUNE_FREQ")) frac = atof(v);
        const double fm = frac * (double)B->n_rows;
        p->freq_min = fm < 1.0 ? 1u : (uint32_t)fm;
    }
    if (st != SG_OK) {
        sg_postings_free(p);
        return st;
    }
    {
        SgTimer timer(ctx, SG_K_POSTINGS);
        SG_HIP_TRY(hipMemsetAsync(p->d_seg, 0, sizeof(uint32_t) * (size_t)(n_bins + 1), ctx->stream));
        SG_HIP_TRY(hipMemsetAsync(cursor, 0, sizeof(uint32_t) * (size_t)(n_bins + 1), ctx->stream));
        // one thread per slot of the (tile x row-in-tile) grid: covers every row, see row_of_thread
        const unsigned grid = (unsigned)((((int64_t)p->n_tiles << tile_log2) + 255) / 256);
        // the tile's counters fit in LDS: one workgroup per tile, LDS atomics (otherwise global ones)
        const size_t lds = (size_t)B->n_cols * 4;
        bool in_lds = B->n_rows > 0 && lds <= 120 * 1024 && lds > 0;
        if (const char *e = getenv("SG_POSTINGS_LDS")) in_lds = in_lds && e[0] != '0';
        const float inv_norm = p->d_filt ? 1.0f / p->norm_up : 0.f;
        if (in_lds) {
            static bool attr_done = false;
            if (!attr_done) {
                (void)hipFuncSetAttribute((const void *)postings_count_lds<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
                (void)hipFuncSetAttribute((const void *)postings_count_lds<double>, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
                (void)hipFuncSetAttribute((const void *)postings_fill_lds<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
                (void)hipFuncSetAttribute((const void *)postings_fill_lds<double>, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
                attr_done = true;
            }
            if (B->dtype == SG_F64)
                hipLaunchKernelGGL(postings_count_lds<double>, dim3((unsigned)p->n_tiles), dim3(1024), lds, ctx->stream, B->d_indptr,
                                   B->d_indices, B->n_rows, tile_log2, p->n_tiles, (int32_t)B->n_cols, p->d_seg);
            else
                hipLaunchKernelGGL(postings_count_lds<float>, dim3((unsigned)p->n_tiles), dim3(1024), lds, ctx->stream, B->d_indptr,
                                   B->d_indices, B->n_rows, tile_log2, p->n_tiles, (int32_t)B->n_cols, p->d_seg);
            SG_HIP_TRY(hipGetLastError());
        } else if (grid > 0) {
            if (B->dtype == SG_F64)
                hipLaunchKernelGGL(postings_count<double>, dim3(grid), dim3(256), 0, ctx->stream, B->d_indptr,
                                   B->d_indices, B->n_rows, tile_log2, p->n_tiles, p->d_seg);
            else
                hipLaunchKernelGGL(postings_count<float>, dim3(grid), dim3(256), 0, ctx->stream, B->d_indptr,
                                   B->d_indices, B->n_rows, tile_log2, p->n_tiles, p->d_seg);
            SG_HIP_TRY(hipGetLastError());
        }
        // counts -> offsets, in place; seg[n_bins] receives the total (= nnz)
        st = sg_exclusive_scan_u32(ctx, p->d_seg, p->d_seg, n_bins, p->d_seg + n_bins);
        if (st == SG_OK && in_lds) {
            if (B->dtype == SG_F64)
                hipLaunchKernelGGL(postings_fill_lds<double>, dim3((unsigned)p->n_tiles), dim3(1024), lds, ctx->stream, B->d_indptr,
                                   B->d_indices, (const double *)B->d_data, B->n_rows, tile_log2, p->n_tiles, (int32_t)B->n_cols,
                                   p->d_seg, p->d_rows, (double *)p->d_vals, p->d_filt, p->freq_min, inv_norm);
            else
                hipLaunchKernelGGL(postings_fill_lds<float>, dim3((unsigned)p->n_tiles), dim3(1024), lds, ctx->stream, B->d_indptr,
                                   B->d_indices, (const float *)B->d_data, B->n_rows, tile_log2, p->n_tiles, (int32_t)B->n_cols,
                                   p->d_seg, p->d_rows, (float *)p->d_vals, p->d_filt, p->freq_min, inv_norm);
            if (hipGetLastError() != hipSuccess) st = SG_ERR_HIP;
        } else if (st == SG_OK && grid > 0) {
            if (B->dtype == SG_F64)
                hipLaunchKernelGGL(postings_fill<double>, dim3(grid), dim3(256), 0, ctx->stream, B->d_indptr,
                                   B->d_indices, (const double *)B->d_data, B->n_rows, tile_log2, p->n_tiles,
                                   p->d_seg, cursor, p->d_rows, (double *)p->d_vals, p->d_filt, p->freq_min, inv_norm);
            else
                hipLaunchKernelGGL(postings_fill<float>, dim3(grid), dim3(256), 0, ctx->stream, B->d_indptr,
                                   B->d_indices, (const float *)B->d_data, B->n_rows, tile_log2, p->n_tiles,
                                   p->d_seg, cursor, p->d_rows, (float *)p->d_vals, p->d_filt, p->freq_min, inv_norm);
            if (hipGetLastError() != hipSuccess) st = SG_ERR_HIP;
        }
        if (st == SG_OK && p->d_ends && B->n_cols > 0) {
            const int64_t cells = (B->n_cols + 1) * (int64_t)p->nt_pad;
            hipLaunchKernelGGL(pack_ends_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, ctx->stream,
                               (const uint32_t *)p->d_seg, B->n_cols, p->n_tiles, p->nt_pad, p->d_ends);
            if (hipGetLastError() != hipSuccess) st = SG_ERR_HIP;
        }
        if (st == SG_OK && p->d_fwd) {
            // the exact scoring reads packed rows in rounds of eight entries and multiplies the slots past a row's end by
            // a = 0: the pad behind the LAST row must hold finite values (0 * NaN would poison that row's score)
            const size_t es = B->dtype == SG_F64 ? 16 : 8;
            if (hipMemsetAsync((char *)p->d_fwd + (size_t)B->nnz * es, 0, 8 * es, ctx->stream) != hipSuccess) st = SG_ERR_HIP;
        }
        if (st == SG_OK && p->d_fwd) {
            const unsigned g2 = (unsigned)((B->n_rows + 1 + 255) / 256);
            if (B->dtype == SG_F64)
                hipLaunchKernelGGL(fwd_pack<double>, dim3(g2), dim3(256), 0, ctx->stream, B->d_indptr, B->d_indices,
                                   (const double *)B->d_data, B->n_rows, p->d_fwd_ptr, p->d_fwd);
            else
                hipLaunchKernelGGL(fwd_pack<float>, dim3(g2), dim3(256), 0, ctx->stream, B->d_indptr, B->d_indices,
                                   (const float *)B->d_data, B->n_rows, p->d_fwd_ptr, p->d_fwd);
            if (hipGetLastError() != hipSuccess) st = SG_ERR_HIP;
        }
    }
    ctx->release(cursor);
    if (st != SG_OK) {
        sg_postings_free(p);
        return st;
    }
    *out = p;
    return SG_OK;
}

extern "C" int sg_postings_free(sg_postings *p) {
    if (!p) return SG_OK;
    p->ctx->release(p->d_seg);
    p->ctx->release(p->d_rows);
    p->ctx->release(p->d_vals);
    p->ctx->release(p->d_fwd);
    p->ctx->release(p->d_fwd_ptr);
    p->ctx->release(p->d_filt);
    p->ctx->release(p->d_ends);
    delete p;
    return SG_OK;
}
