// K6 -- self-join post-processing of fit() on the device.
//
// Replaces the scipy lil round trip of the reference:
//   matches.tolil(); _fix_diagonal (m[r, r] = 1 for every row)      string_grouper.py:419-424, :954-958
//   _symmetrize_matrix (m[c, r] = m[r, c] for every stored (r, c))   string_grouper.py:425-427, :960-964
//   _get_matches_list (CSR -> (master_side, dupe_side, similarity))   string_grouper.py:755-763
// Input: the fixed-stride top-n result of K4 (square).  Output: a CSR-ordered match list, rows sorted
// by column like the reference's lil -> csr conversion leaves them.
//
// A stored pair and its mirror carry bit-identical scores (same ascending-k order, commutative
// products), so "mirror if absent" is all the symmetrisation has to do.  Three passes over the
// (at most n * top_n) entries + a per-row sort of the (short) rows: HBM bound, a few ms at 663 k.
#include "sg_internal.h"

struct sg_matchlist {
    sg_ctx *ctx = nullptr;
    int64_t n_rows = 0, n_cols = 0, n_entries = 0;
    int32_t dtype = SG_F32;
    int64_t *d_row_ptr = nullptr;   // n_rows + 1
    int32_t *d_cols = nullptr;      // n_entries, ascending inside a row
    void *d_vals = nullptr;
};

// for the reductions over the list (sg_reduce.hip)
int sg_matchlist_device_view(const sg_matchlist *ml, int64_t *n_rows, int64_t *n_cols, int64_t *n_entries, int32_t *dtype,
                             const int64_t **row_ptr, const int32_t **cols, const void **vals) {
    SG_REQUIRE(ml != nullptr, "match list is null");
    *n_rows = ml->n_rows;
    *n_cols = ml->n_cols;
    *n_entries = ml->n_entries;
    *dtype = ml->dtype;
    *row_ptr = ml->d_row_ptr;
    *cols = ml->d_cols;
    *vals = ml->d_vals;
    return SG_OK;
}

__device__ __forceinline__ bool row_has(const int32_t *cols, const int32_t *cnt, int32_t stride, int64_t row, int32_t col) {
    const int32_t *rc = cols + (size_t)row * stride;
    const int n = cnt[row];
    for (int q = 0; q < n; ++q)
        if (rc[q] == col) return true;
    return false;
}

// own[r] = entries row r keeps itself (off-diagonal ones + the diagonal), extra[c] += 1 for every stored
// (r, c) whose mirror (c, r) is not stored
__global__ void __launch_bounds__(256) ml_count_kernel(const int32_t *__restrict__ cols, const int32_t *__restrict__ cnt,
                                                       int32_t stride, int64_t n, int fix_diag, int symmetrize,
                                                       int32_t *own, int32_t *extra) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const int32_t *rc = cols + (size_t)r * stride;
    const int m = cnt[r];
    int keep = 0;
    bool diag = false;
    for (int e = 0; e < m; ++e) {
        const int32_t c = rc[e];
        if (c == (int32_t)r) {
            diag = true;
            continue;
        }
        ++keep;
        if (symmetrize && c < n && !row_has(cols, cnt, stride, c, (int32_t)r)) atomicAdd(&extra[c], 1);
    }
    own[r] = keep + ((fix_diag || diag) ? 1 : 0);
}

template <typename T>
__global__ void __launch_bounds__(256) ml_fill_kernel(const int32_t *__restrict__ cols, const T *__restrict__ vals,
                                                      const int32_t *__restrict__ cnt, int32_t stride, int64_t n,
                                                      int fix_diag, int symmetrize, const int64_t *__restrict__ row_ptr,
                                                      const int32_t *__restrict__ own, int32_t *cursor,
                                                      int32_t *out_cols, T *out_vals) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const int32_t *rc = cols + (size_t)r * stride;
    const T *rv = vals + (size_t)r * stride;
    const int m = cnt[r];
    int64_t o = row_ptr[r];
    bool diag = false;
    T diag_val = (T)1;
    for (int e = 0; e < m; ++e) {
        const int32_t c = rc[e];
        if (c == (int32_t)r) {
            diag = true;
            diag_val = rv[e];
            continue;
        }
        out_cols[o] = c;
        out_vals[o] = rv[e];
        ++o;
        if (symmetrize && c < n && !row_has(cols, cnt, stride, c, (int32_t)r)) {
            const int64_t p = row_ptr[c] + own[c] + atomicAdd(&cursor[c], 1);
            out_cols[p] = (int32_t)r;
            out_vals[p] = rv[e];
        }
    }
    if (fix_diag || diag) {
        out_cols[o] = (int32_t)r;
        out_vals[o] = fix_diag ? (T)1 : diag_val;
    }
}

// Order every row by column.  One wave per row; rank by counting (columns of a row are distinct), which
// is O(len^2 / 64) but rows hold a handful of entries -- only "hub" strings that many others list have
// long ones.
template <typename T>
__global__ void __launch_bounds__(64) ml_sort_rows_kernel(const int64_t *__restrict__ row_ptr, int64_t n,
                                                          const int32_t *__restrict__ in_cols,
                                                          const T *__restrict__ in_vals, int32_t *out_cols, T *out_vals) {
    const int lane = threadIdx.x;
    for (int64_t r = blockIdx.x; r < n; r += gridDim.x) {
        const int64_t lo = row_ptr[r];
        const int64_t len = row_ptr[r + 1] - lo;
        for (int64_t i = lane; i < len; i += 64) {
            const int32_t c = in_cols[lo + i];
            int64_t rank = 0;
            for (int64_t q = 0; q < len; ++q) rank += in_cols[lo + q] < c;
            out_cols[lo + rank] = c;
            out_vals[lo + rank] = in_vals[lo + i];
        }
    }
}

__global__ void __launch_bounds__(256) ml_add_kernel(const int32_t *a, const int32_t *b, int32_t *c, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) c[i] = a[i] + b[i];
}

// plain compaction of a fixed-stride result (two-series case: order inside a row is kept)
template <typename T>
__global__ void __launch_bounds__(256) ml_compact_kernel(const int32_t *__restrict__ cols, const T *__restrict__ vals,
                                                         const int32_t *__restrict__ cnt, int32_t stride, int64_t n,
                                                         const int64_t *__restrict__ row_ptr, int32_t *out_cols,
                                                         T *out_vals) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const int64_t o = row_ptr[r];
    for (int e = 0; e < cnt[r]; ++e) {
        out_cols[o + e] = cols[(size_t)r * stride + e];
        out_vals[o + e] = vals[(size_t)r * stride + e];
    }
}

extern "C" int sg_matchlist_free(sg_matchlist *ml) {
    if (!ml) return SG_OK;
    ml->ctx->release(ml->d_row_ptr);
    ml->ctx->release(ml->d_cols);
    ml->ctx->release(ml->d_vals);
    delete ml;
    return SG_OK;
}

extern "C" int sg_matchlist_build(sg_ctx *ctx, const sg_topn *r, int32_t fix_diagonal, int32_t symmetrize,
                                  int32_t sort_by_column, sg_matchlist **out) {
    SG_REQUIRE(ctx && r && out, "null argument");
    if (fix_diagonal || symmetrize) SG_REQUIRE(r->n_rows == r->n_cols, "self-join post-processing needs a square result");
    const int64_t n = r->n_rows;
    sg_matchlist *ml = new (std::nothrow) sg_matchlist();
    if (!ml) return SG_ERR_OOM;
    ml->ctx = ctx;
    ml->n_rows = n;
    ml->n_cols = r->n_cols;
    ml->dtype = r->dtype;
    const size_t s = r->dtype == SG_F64 ? 8 : 4;
    int32_t *own = nullptr, *extra = nullptr, *total = nullptr, *tmp_cols = nullptr;
    void *tmp_vals = nullptr;
    int st = sg_alloc(ctx, (size_t)n + 1, &own);
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)n + 1, &extra);
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)n + 1, &total);
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)n + 2, &ml->d_row_ptr);
    const unsigned grid = (unsigned)((n + 255) / 256);
    const bool self = fix_diagonal || symmetrize;
    if (st == SG_OK && n > 0) {
        (void)hipMemsetAsync(extra, 0, sizeof(int32_t) * (size_t)(n + 1), ctx->stream);
        if (self) {
            hipLaunchKernelGGL(ml_count_kernel, dim3(grid), dim3(256), 0, ctx->stream, (const int32_t *)r->d_cols,
                               (const int32_t *)r->d_counts, r->stride, n, fix_diagonal, symmetrize, own, extra);
        } else {
            (void)hipMemcpyAsync(own, r->d_counts, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToDevice, ctx->stream);
        }
    }
    if (st == SG_OK && n > 0) {
        hipLaunchKernelGGL(ml_add_kernel, dim3(grid), dim3(256), 0, ctx->stream, (const int32_t *)own,
                           (const int32_t *)extra, total, n);
        st = sg_exclusive_scan_i32_to_i64(ctx, total, ml->d_row_ptr, n);
    } else if (st == SG_OK) {
        (void)hipMemsetAsync(ml->d_row_ptr, 0, sizeof(int64_t), ctx->stream);
    }
    int64_t n_entries = 0;
    if (st == SG_OK) {
        if (hipMemcpyAsync(&n_entries, ml->d_row_ptr + n, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess)
            st = SG_ERR_HIP;
    }
    ml->n_entries = n_entries;
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)n_entries + 4, &ml->d_cols);
    if (st == SG_OK) st = ctx->alloc(((size_t)n_entries + 4) * s, &ml->d_vals);
    const bool sorted = self || sort_by_column;
    if (st == SG_OK && sorted) st = sg_alloc(ctx, (size_t)n_entries + 4, &tmp_cols);
    if (st == SG_OK && sorted) st = ctx->alloc(((size_t)n_entries + 4) * s, &tmp_vals);
    if (st == SG_OK && n > 0) {
        if (self) {
            (void)hipMemsetAsync(extra, 0, sizeof(int32_t) * (size_t)(n + 1), ctx->stream);   // now the mirror cursors
            const unsigned sgrid = (unsigned)(n < 256 * 64 ? n : 256 * 64);
            if (r->dtype == SG_F64) {
                hipLaunchKernelGGL(ml_fill_kernel<double>, dim3(grid), dim3(256), 0, ctx->stream, (const int32_t *)r->d_cols,
                                   (const double *)r->d_vals, (const int32_t *)r->d_counts, r->stride, n, fix_diagonal,
                                   symmetrize, (const int64_t *)ml->d_row_ptr, (const int32_t *)own, extra, tmp_cols,
                                   (double *)tmp_vals);
                hipLaunchKernelGGL(ml_sort_rows_kernel<double>, dim3(sgrid), dim3(64), 0, ctx->stream,
                                   (const int64_t *)ml->d_row_ptr, n, (const int32_t *)tmp_cols, (const double *)tmp_vals,
                                   ml->d_cols, (double *)ml->d_vals);
            } else {
                hipLaunchKernelGGL(ml_fill_kernel<float>, dim3(grid), dim3(256), 0, ctx->stream, (const int32_t *)r->d_cols,
                                   (const float *)r->d_vals, (const int32_t *)r->d_counts, r->stride, n, fix_diagonal,
                                   symmetrize, (const int64_t *)ml->d_row_ptr, (const int32_t *)own, extra, tmp_cols,
                                   (float *)tmp_vals);
                hipLaunchKernelGGL(ml_sort_rows_kernel<float>, dim3(sgrid), dim3(64), 0, ctx->stream,
                                   (const int64_t *)ml->d_row_ptr, n, (const int32_t *)tmp_cols, (const float *)tmp_vals,
                                   ml->d_cols, (float *)ml->d_vals);
            }
        } else {
            // plain compaction; optionally followed by the per-row column sort
            int32_t *c_out = sort_by_column ? tmp_cols : ml->d_cols;
            void *v_out = sort_by_column ? tmp_vals : ml->d_vals;
            const unsigned sgrid = (unsigned)(n < 256 * 64 ? n : 256 * 64);
            if (r->dtype == SG_F64) {
                hipLaunchKernelGGL(ml_compact_kernel<double>, dim3(grid), dim3(256), 0, ctx->stream,
                                   (const int32_t *)r->d_cols, (const double *)r->d_vals, (const int32_t *)r->d_counts,
                                   r->stride, n, (const int64_t *)ml->d_row_ptr, c_out, (double *)v_out);
                if (sort_by_column)
                    hipLaunchKernelGGL(ml_sort_rows_kernel<double>, dim3(sgrid), dim3(64), 0, ctx->stream,
                                       (const int64_t *)ml->d_row_ptr, n, (const int32_t *)tmp_cols,
                                       (const double *)tmp_vals, ml->d_cols, (double *)ml->d_vals);
            } else {
                hipLaunchKernelGGL(ml_compact_kernel<float>, dim3(grid), dim3(256), 0, ctx->stream,
                                   (const int32_t *)r->d_cols, (const float *)r->d_vals, (const int32_t *)r->d_counts,
                                   r->stride, n, (const int64_t *)ml->d_row_ptr, c_out, (float *)v_out);
                if (sort_by_column)
                    hipLaunchKernelGGL(ml_sort_rows_kernel<float>, dim3(sgrid), dim3(64), 0, ctx->stream,
                                       (const int64_t *)ml->d_row_ptr, n, (const int32_t *)tmp_cols,
                                       (const float *)tmp_vals, ml->d_cols, (float *)ml->d_vals);
            }
        }
        if (hipGetLastError() != hipSuccess) st = SG_ERR_HIP;
    }
    ctx->release(own);
    ctx->release(extra);
    ctx->release(total);
    ctx->release(tmp_cols);
    ctx->release(tmp_vals);
    if (st != SG_OK) {
        sg_matchlist_free(ml);
        return st;
    }
    *out = ml;
    return SG_OK;
}

extern "C" int sg_matchlist_dims(const sg_matchlist *ml, int64_t *n_rows, int64_t *n_entries, int32_t *dtype) {
    SG_REQUIRE(ml != nullptr, "match list is null");
    if (n_rows) *n_rows = ml->n_rows;
    if (n_entries) *n_entries = ml->n_entries;
    if (dtype) *dtype = ml->dtype;
    return SG_OK;
}

extern "C" int sg_matchlist_to_host(sg_ctx *ctx, const sg_matchlist *ml, int64_t *row_ptr, int32_t *cols, void *vals) {
    SG_REQUIRE(ctx && ml && row_ptr, "null argument");
    const size_t s = ml->dtype == SG_F64 ? 8 : 4;
    SG_HIP_TRY(hipMemcpyAsync(row_ptr, ml->d_row_ptr, sizeof(int64_t) * (size_t)(ml->n_rows + 1), hipMemcpyDeviceToHost,
                              ctx->stream));
    if (ml->n_entries > 0) {
        SG_REQUIRE(cols && vals, "null output");
        SG_HIP_TRY(hipMemcpyAsync(cols, ml->d_cols, sizeof(int32_t) * (size_t)ml->n_entries, hipMemcpyDeviceToHost, ctx->stream));
        SG_HIP_TRY(hipMemcpyAsync(vals, ml->d_vals, s * (size_t)ml->n_entries, hipMemcpyDeviceToHost, ctx->stream));
    }
    SG_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SG_OK;
}
