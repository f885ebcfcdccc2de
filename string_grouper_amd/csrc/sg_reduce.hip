// K7 / K8 -- the two reductions the reference runs over the match list, on the device
// (SURVEY.md section 8f, rows f3 and f2); K9 -- the row-wise similarity of dot() (row f4).  Both read the CSR-ordered match list K6 leaves in HBM and
// return ONE int32 per string instead of the list itself.
//
// K7  best master per duplicate (match_most_similar):                string_grouper.py:803-807
//       dupes_max_sim = matches_list.groupby('dupe_side').agg({'similarity': 'max'}) ... merge ...
//       .groupby(['dupe_side']).agg({'master_side': 'min'})
//     = for every column c the row r with the largest similarity, the lowest r among equals.
// K8  group representative per string (group_similar_strings):       string_grouper.py:851-904
//       connected_components(csgraph=graph, directed=True)   (weak connectivity, scipy default)
//       group_rep='first'    -> the member with the lowest index
//       group_rep='centroid' -> the member with the largest row sum of similarities
//                               (graph.sum(axis=1): float64, numpy's pairwise order), the lowest
//                               index among equals
//
// Similarities are > 0 (they passed a threshold >= 0), so their IEEE bit patterns order like unsigned
// integers and "max similarity, then min row" is two integer atomics per entry.  Connected components:
// minimum-label propagation with hooking and pointer jumping (every label is the index of a member of
// the same component, so the fixed point labels each component with its lowest index -- which IS the
// 'first' representative).  A few HBM passes over <= n * top_n entries each: microseconds to
// milliseconds, against seconds for the pandas group-bys / scipy csgraph at 10^7 entries.
#include "sg_internal.h"

template <typename T>
struct Bits;
template <>
struct Bits<float> {
    typedef unsigned int type;
    static __device__ __forceinline__ type of(float v) { return __float_as_uint(v); }
};
template <>
struct Bits<double> {
    typedef unsigned long long type;
    static __device__ __forceinline__ type of(double v) { return (unsigned long long)__double_as_longlong(v); }
};

// ---------------------------------------------------------------------------------------------- K7
template <typename T>
__global__ void __launch_bounds__(256) best_max_kernel(const int64_t *__restrict__ row_ptr, const int32_t *__restrict__ cols,
                                                       const T *__restrict__ vals, int64_t n_rows,
                                                       typename Bits<T>::type *col_max) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    for (int64_t p = row_ptr[r]; p < row_ptr[r + 1]; ++p) atomicMax(&col_max[cols[p]], Bits<T>::of(vals[p]));
}

template <typename T>
__global__ void __launch_bounds__(256) best_row_kernel(const int64_t *__restrict__ row_ptr, const int32_t *__restrict__ cols,
                                                       const T *__restrict__ vals, int64_t n_rows,
                                                       const typename Bits<T>::type *__restrict__ col_max, int32_t *best) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    for (int64_t p = row_ptr[r]; p < row_ptr[r + 1]; ++p) {
        const int32_t c = cols[p];
        if (Bits<T>::of(vals[p]) == col_max[c]) atomicMin(&best[c], (int32_t)r);
    }
}

__global__ void __launch_bounds__(256) best_finish_kernel(int32_t *best, int64_t n) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c < n && best[c] == INT32_MAX) best[c] = -1;   // no match for this duplicate
}

// ---------------------------------------------------------------------------------------------- K8
__global__ void __launch_bounds__(256) cc_init_kernel(int32_t *label, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) label[i] = (int32_t)i;
}

// one round: every edge (r, c) pulls both ends (and their current roots) down to the smaller label
__global__ void __launch_bounds__(256) cc_hook_kernel(const int64_t *__restrict__ row_ptr, const int32_t *__restrict__ cols,
                                                      int64_t n_rows, int32_t *label, int32_t *changed) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    bool any = false;
    for (int64_t p = row_ptr[r]; p < row_ptr[r + 1]; ++p) {
        const int32_t c = cols[p];
        if (c < 0 || c >= n_rows) continue;
        const int32_t lr = label[r], lc = label[c];
        if (lr == lc) continue;
        const int32_t lo = lr < lc ? lr : lc, hi = lr < lc ? lc : lr;
        atomicMin(&label[hi], lo);          // hook the larger root under the smaller label
        atomicMin(&label[lr < lc ? c : (int32_t)r], lo);
        any = true;
    }
    if (any) *changed = 1;
}

__global__ void __launch_bounds__(256) cc_jump_kernel(int32_t *label, int64_t n, int32_t *changed) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int32_t l = label[i];
    int32_t ll = label[l];
    bool any = false;
    while (ll != l) {   // follow the chain to its current root
        l = ll;
        ll = label[l];
        any = true;
    }
    if (any) {
        label[i] = l;
        *changed = 1;
    }
}

// weight[i] = row sum of similarities in float64 -- the match list's similarity column is float64
// whatever the TF-IDF dtype (string_grouper.py:750 up-casts) -- in the order the reference gets it:
// graph.sum(axis=1) is np.add.reduceat over the row (scipy _minor_reduce), i.e. the first element plus
// numpy's pairwise sum of the rest (numpy/_core/src/umath/loops_utils.h.src, @TYPE@_pairwise_sum:
// fewer than 8 elements sequentially; up to 128 with eight strided partial sums combined as
// ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) and the tail added sequentially; longer runs split in halves
// rounded down to a multiple of 8).  The centroid is an arg-max over these sums, so the bits matter.
template <typename ACC, typename T>   // ACC: the type numpy accumulates in (the array's own: float for float32)
__device__ ACC np_pairwise_sum(const T *a, int64_t n) {
    if (n < 8) {
        ACC res = (ACC)0;
        for (int64_t i = 0; i < n; ++i) res = res + (ACC)a[i];
        return res;
    }
    if (n <= 128) {
        ACC r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = (ACC)a[j];
        int64_t i = 8;
        for (; i < n - (n % 8); i += 8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] = r[j] + (ACC)a[i + j];
        }
        ACC res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res = res + (ACC)a[i];
        return res;
    }
    int64_t n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise_sum<ACC, T>(a, n2) + np_pairwise_sum<ACC, T>(a + n2, n - n2);
}

template <typename T>
__global__ void __launch_bounds__(256) row_weight_kernel(const int64_t *__restrict__ row_ptr, const T *__restrict__ vals,
                                                         int64_t n_rows, double *weight) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    const int64_t lo = row_ptr[r], m = row_ptr[r + 1] - lo;
    weight[r] = m == 0 ? 0.0 : (double)vals[lo] + np_pairwise_sum<double, T>(vals + lo + 1, m - 1);
}

template <typename T>
__global__ void __launch_bounds__(256) rep_max_kernel(const int32_t *__restrict__ label, const T *__restrict__ weight,
                                                      int64_t n, typename Bits<T>::type *group_max) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) atomicMax(&group_max[label[i]], Bits<T>::of(weight[i]));
}

template <typename T>
__global__ void __launch_bounds__(256) rep_pick_kernel(const int32_t *__restrict__ label, const T *__restrict__ weight,
                                                       int64_t n, const typename Bits<T>::type *__restrict__ group_max,
                                                       int32_t *group_rep) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && Bits<T>::of(weight[i]) == group_max[label[i]]) atomicMin(&group_rep[label[i]], (int32_t)i);
}

__global__ void __launch_bounds__(256) rep_gather_kernel(const int32_t *__restrict__ label,
                                                         const int32_t *__restrict__ group_rep, int64_t n, int32_t *out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = group_rep[label[i]];
}

// ---------------------------------------------------------------------------------------------- K9
// Row-wise similarity of two equally shaped matrices (StringGrouper.dot, string_grouper.py:433-440):
//   np.asarray(master_matrix.multiply(duplicate_matrix).sum(axis=1))
// scipy's element-wise product keeps the non-zero products of the common columns in ascending column
// order (csr_binop_csr_canonical), rounded to T; the row sum is again np.add.reduceat: first product +
// numpy's pairwise sum of the rest, accumulated in T.  One thread per row: the products go to a scratch
// segment (at most the row's length in A), then the sum is taken from there.
template <typename T>
__global__ void __launch_bounds__(256) rowwise_dot_kernel(const int64_t *__restrict__ a_indptr, const int32_t *__restrict__ a_indices,
                                                          const T *__restrict__ a_data, const int64_t *__restrict__ b_indptr,
                                                          const int32_t *__restrict__ b_indices, const T *__restrict__ b_data,
                                                          int64_t n_rows, T *scratch, T *out) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    int64_t pa = a_indptr[r], pb = b_indptr[r];
    const int64_t ea = a_indptr[r + 1], eb = b_indptr[r + 1];
    T *prod = scratch + (a_indptr[r] - a_indptr[0]);
    int64_t m = 0;
    while (pa < ea && pb < eb) {
        const int32_t ka = a_indices[pa], kb = b_indices[pb];
        if (ka == kb) {
            const T p = a_data[pa] * b_data[pb];
            if (p != (T)0) prod[m++] = p;
            ++pa;
            ++pb;
        } else if (ka < kb) {
            ++pa;
        } else {
            ++pb;
        }
    }
    out[r] = m == 0 ? (T)0 : prod[0] + np_pairwise_sum<T, T>(prod + 1, m - 1);
}

// ---------------------------------------------------------------------------------------------- host
static inline unsigned blocks_for(int64_t n) { return (unsigned)((n + 255) / 256); }

extern "C" int sg_matchlist_best_master(sg_ctx *ctx, const sg_matchlist *ml, int32_t *out_best) {
    SG_REQUIRE(ctx && ml && out_best, "null argument");
    int64_t n_rows = 0, n_cols = 0, n_entries = 0;
    int32_t dtype = 0;
    const int64_t *row_ptr = nullptr;
    const int32_t *cols = nullptr;
    const void *vals = nullptr;
    SG_TRY(sg_matchlist_device_view(ml, &n_rows, &n_cols, &n_entries, &dtype, &row_ptr, &cols, &vals));
    if (n_cols == 0) return SG_OK;
    const size_t bs = dtype == SG_F64 ? 8 : 4;
    void *col_max = nullptr;
    int32_t *best = nullptr;
    int st = ctx->alloc((size_t)n_cols * bs, &col_max);
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)n_cols, &best);
    hipError_t e = hipSuccess;
    if (st == SG_OK) e = hipMemsetAsync(col_max, 0, (size_t)n_cols * bs, ctx->stream);
    if (st == SG_OK && e == hipSuccess) {
        e = hipMemsetD32Async((hipDeviceptr_t)best, INT32_MAX, (size_t)n_cols, ctx->stream);
        if (e == hipSuccess && n_rows > 0) {
            if (dtype == SG_F64) {
                hipLaunchKernelGGL(best_max_kernel<double>, dim3(blocks_for(n_rows)), dim3(256), 0, ctx->stream, row_ptr, cols,
                                   (const double *)vals, n_rows, (unsigned long long *)col_max);
                hipLaunchKernelGGL(best_row_kernel<double>, dim3(blocks_for(n_rows)), dim3(256), 0, ctx->stream, row_ptr, cols,
                                   (const double *)vals, n_rows, (const unsigned long long *)col_max, best);
            } else {
                hipLaunchKernelGGL(best_max_kernel<float>, dim3(blocks_for(n_rows)), dim3(256), 0, ctx->stream, row_ptr, cols,
                                   (const float *)vals, n_rows, (unsigned int *)col_max);
                hipLaunchKernelGGL(best_row_kernel<float>, dim3(blocks_for(n_rows)), dim3(256), 0, ctx->stream, row_ptr, cols,
                                   (const float *)vals, n_rows, (const unsigned int *)col_max, best);
            }
        }
        if (e == hipSuccess) {
            hipLaunchKernelGGL(best_finish_kernel, dim3(blocks_for(n_cols)), dim3(256), 0, ctx->stream, best, n_cols);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(out_best, best, (size_t)n_cols * 4, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    }
    ctx->release(col_max);
    ctx->release(best);
    if (st != SG_OK) return st;
    if (e != hipSuccess) {
        sg_set_error("sg_matchlist_best_master: %s", hipGetErrorString(e));
        return SG_ERR_HIP;
    }
    return SG_OK;
}

extern "C" int sg_matchlist_group_reps(sg_ctx *ctx, const sg_matchlist *ml, int32_t centroid, int32_t *out_rep) {
    SG_REQUIRE(ctx && ml && out_rep, "null argument");
    int64_t n = 0, n_cols = 0, n_entries = 0;
    int32_t dtype = 0;
    const int64_t *row_ptr = nullptr;
    const int32_t *cols = nullptr;
    const void *vals = nullptr;
    SG_TRY(sg_matchlist_device_view(ml, &n, &n_cols, &n_entries, &dtype, &row_ptr, &cols, &vals));
    SG_REQUIRE(n == n_cols, "group representatives need a square match list (a self-join)");
    if (n == 0) return SG_OK;
    int32_t *label = nullptr, *flag = nullptr, *rep = nullptr, *out = nullptr;
    void *weight = nullptr, *gmax = nullptr;
    int st = sg_alloc(ctx, (size_t)n, &label);
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)16, &flag);
    if (st == SG_OK && centroid) {
        st = sg_alloc(ctx, (size_t)n, &rep);
        if (st == SG_OK) st = sg_alloc(ctx, (size_t)n, &out);
        if (st == SG_OK) st = ctx->alloc((size_t)n * 8, &weight);
        if (st == SG_OK) st = ctx->alloc((size_t)n * 8, &gmax);
    }
    hipError_t e = hipSuccess;
    const unsigned grid = blocks_for(n);
    if (st == SG_OK) {
        hipLaunchKernelGGL(cc_init_kernel, dim3(grid), dim3(256), 0, ctx->stream, label, n);
        int32_t changed = 1;
        int rounds = 0;
        while (changed && e == hipSuccess) {
            if (++rounds > 100000) {
                sg_set_error("connected components did not converge");
                st = SG_ERR_HIP;
                break;
            }
            e = hipMemsetAsync(flag, 0, 4, ctx->stream);
            hipLaunchKernelGGL(cc_hook_kernel, dim3(grid), dim3(256), 0, ctx->stream, row_ptr, cols, n, label, flag);
            hipLaunchKernelGGL(cc_jump_kernel, dim3(grid), dim3(256), 0, ctx->stream, label, n, flag);
            if (e == hipSuccess) e = hipMemcpyAsync(&changed, flag, 4, hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        }
    }
    const int32_t *result = label;   // 'first': the label is the lowest index of the component
    if (st == SG_OK && e == hipSuccess && centroid) {
        e = hipMemsetAsync(gmax, 0, (size_t)n * 8, ctx->stream);
        if (e == hipSuccess) e = hipMemsetD32Async((hipDeviceptr_t)rep, INT32_MAX, (size_t)n, ctx->stream);
        if (e == hipSuccess) {
            if (dtype == SG_F64)
                hipLaunchKernelGGL(row_weight_kernel<double>, dim3(grid), dim3(256), 0, ctx->stream, row_ptr, (const double *)vals, n,
                                   (double *)weight);
            else
                hipLaunchKernelGGL(row_weight_kernel<float>, dim3(grid), dim3(256), 0, ctx->stream, row_ptr, (const float *)vals, n,
                                   (double *)weight);
            hipLaunchKernelGGL(rep_max_kernel<double>, dim3(grid), dim3(256), 0, ctx->stream, label, (const double *)weight, n,
                               (unsigned long long *)gmax);
            hipLaunchKernelGGL(rep_pick_kernel<double>, dim3(grid), dim3(256), 0, ctx->stream, label, (const double *)weight, n,
                               (const unsigned long long *)gmax, rep);
            hipLaunchKernelGGL(rep_gather_kernel, dim3(grid), dim3(256), 0, ctx->stream, label, rep, n, out);
            e = hipGetLastError();
            result = out;
        }
    }
    if (st == SG_OK && e == hipSuccess) {
        e = hipMemcpyAsync(out_rep, result, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    }
    for (void *p : {(void *)label, (void *)flag, (void *)rep, (void *)out, weight, gmax}) ctx->release(p);
    if (st != SG_OK) return st;
    if (e != hipSuccess) {
        sg_set_error("sg_matchlist_group_reps: %s", hipGetErrorString(e));
        return SG_ERR_HIP;
    }
    return SG_OK;
}

extern "C" int sg_csr_rowwise_dot(sg_ctx *ctx, const sg_csr *A, const sg_csr *B, void *out_host) {
    SG_REQUIRE(ctx && A && B && out_host, "null argument");
    SG_REQUIRE(A->n_rows == B->n_rows && A->n_cols == B->n_cols, "matrices differ in shape");
    SG_REQUIRE(A->dtype == B->dtype, "matrices differ in value type");
    const int64_t n = A->n_rows;
    if (n == 0) return SG_OK;
    const size_t s = A->dtype == SG_F64 ? 8 : 4;
    void *scratch = nullptr, *out = nullptr;
    int st = ctx->alloc(((size_t)A->nnz + 1) * s, &scratch);
    if (st == SG_OK) st = ctx->alloc((size_t)n * s, &out);
    hipError_t e = hipSuccess;
    if (st == SG_OK) {
        if (A->dtype == SG_F64)
            hipLaunchKernelGGL(rowwise_dot_kernel<double>, dim3(blocks_for(n)), dim3(256), 0, ctx->stream, A->d_indptr, A->d_indices,
                               (const double *)A->d_data, B->d_indptr, B->d_indices, (const double *)B->d_data, n,
                               (double *)scratch, (double *)out);
        else
            hipLaunchKernelGGL(rowwise_dot_kernel<float>, dim3(blocks_for(n)), dim3(256), 0, ctx->stream, A->d_indptr, A->d_indices,
                               (const float *)A->d_data, B->d_indptr, B->d_indices, (const float *)B->d_data, n,
                               (float *)scratch, (float *)out);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(out_host, out, (size_t)n * s, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    }
    ctx->release(scratch);
    ctx->release(out);
    if (st != SG_OK) return st;
    if (e != hipSuccess) {
        sg_set_error("sg_csr_rowwise_dot: %s", hipGetErrorString(e));
        return SG_ERR_HIP;
    }
    return SG_OK;
}
