// Sorted vocabulary for n-gram key spaces too wide for a dense table (more than 30 bits: long n-grams, large
// alphabets): all (row, distinct n-gram) keys of the fit are sorted and run-length encoded -- the distinct keys in
// ascending order ARE the vocabulary in sklearn's order (column = rank, text.py:1194-1206), and because a key occurs
// at most once per row its run length is its document frequency.  The device-wide radix sort and run-length encode
// are rocPRIM's (the ROCm primitive library); everything around them is this library's.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "sg_internal.h"

int sg_sort_unique_u64(sg_ctx *ctx, uint64_t *d_keys, int64_t n, uint64_t *d_unique, int32_t *d_counts, int64_t *n_unique) {
    *n_unique = 0;
    if (n <= 0) return SG_OK;
    if (n >= ((int64_t)1 << 31)) {
        sg_set_error("%lld n-gram occurrences exceed the 32-bit run-length encoder; split the input", (long long)n);
        return SG_ERR_OVERFLOW;
    }
    uint64_t *d_sorted = nullptr;
    uint32_t *d_runs = nullptr;
    void *d_tmp = nullptr;
    int st = sg_alloc(ctx, (size_t)n, &d_sorted);
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)4, &d_runs);
    size_t tmp1 = 0, tmp2 = 0;
    hipError_t e = hipSuccess;
    if (st == SG_OK) {
        e = rocprim::radix_sort_keys(nullptr, tmp1, d_keys, d_sorted, (size_t)n, 0, 64, ctx->stream);
        if (e == hipSuccess)
            e = rocprim::run_length_encode(nullptr, tmp2, d_sorted, (unsigned int)n, d_unique, (unsigned int *)d_counts,
                                           d_runs, ctx->stream);
        if (e == hipSuccess) st = ctx->alloc(tmp1 > tmp2 ? tmp1 : tmp2, &d_tmp);
    }
    uint32_t runs = 0;
    if (st == SG_OK && e == hipSuccess) {
        e = rocprim::radix_sort_keys(d_tmp, tmp1, d_keys, d_sorted, (size_t)n, 0, 64, ctx->stream);
        if (e == hipSuccess)
            e = rocprim::run_length_encode(d_tmp, tmp2, d_sorted, (unsigned int)n, d_unique, (unsigned int *)d_counts, d_runs,
                                           ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(&runs, d_runs, 4, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    }
    ctx->release(d_sorted);
    ctx->release(d_runs);
    ctx->release(d_tmp);
    if (st != SG_OK) return st;
    if (e != hipSuccess) {
        sg_set_error("sorting the vocabulary failed: %s", hipGetErrorString(e));
        return SG_ERR_HIP;
    }
    *n_unique = runs;
    return SG_OK;
}

int sg_sort_pairs_u64_u32(sg_ctx *ctx, const uint64_t *d_keys, const uint32_t *d_vals, int64_t n, uint64_t *d_keys_out,
                          uint32_t *d_vals_out) {
    if (n <= 0) return SG_OK;
    size_t tmp = 0;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, tmp, d_keys, d_keys_out, d_vals, d_vals_out, (size_t)n, 0, 64, ctx->stream);
    void *d_tmp = nullptr;
    int st = e == hipSuccess ? ctx->alloc(tmp ? tmp : 256, &d_tmp) : SG_ERR_HIP;
    if (st == SG_OK) {
        e = rocprim::radix_sort_pairs(d_tmp, tmp, d_keys, d_keys_out, d_vals, d_vals_out, (size_t)n, 0, 64, ctx->stream);
        if (e != hipSuccess) st = SG_ERR_HIP;
    }
    ctx->release(d_tmp);   // (stream-ordered reuse)
    if (st != SG_OK) sg_set_error("sorting (hash, row) pairs failed: %s", hipGetErrorString(e));
    return st;
}
