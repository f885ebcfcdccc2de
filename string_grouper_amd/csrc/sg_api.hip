// Context, scratch pool, object plumbing and the prefix-sum utility of libsg_hip.so.
// gfx950 (MI355X) only; no CPU fallback lives in this library.
#include <stdarg.h>
#include <stdlib.h>

#include "sg_internal.h"
#include "sg_scan.h"

// ------------------------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";

void sg_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *sg_last_error(void) { return g_err; }
extern "C" int sg_abi_version(void) { return SG_ABI_VERSION; }

extern "C" int sg_device_count(int *count) {
    SG_REQUIRE(count != nullptr, "count is null");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        n = 0;
    }
    *count = n;
    return SG_OK;
}

// ------------------------------------------------------------------------------------ pool
// SG_POISON_ALLOC=1 (test hook): every block handed out is filled with 0xFF bytes (NaN as a float, -1 as an index)
// first, so that a kernel reading memory it never wrote shows up as a wrong result instead of depending on what
// the pool happened to hold (tests/test_parity_gpu.py::test_results_do_not_depend_on_uninitialised_memory).
extern char **environ;
static void snapshot_options(sg_ctx *ctx) {
    ctx->opts.clear();
    for (char **e = environ; e && *e; ++e) {
        if (strncmp(*e, "SG_", 3) != 0) continue;
        const char *eq = strchr(*e, '=');
        if (!eq) continue;
        ctx->opts[std::string(*e, (size_t)(eq - *e))] = std::string(eq + 1);
    }
    const char *v = ctx->opt("SG_POISON_ALLOC");
    ctx->poison = v && v[0] == '1';
}

int sg_ctx::alloc(size_t bytes, void **out) {
    if (bytes == 0) bytes = 256;
    bytes = (bytes + 255) & ~(size_t)255;
    {
        std::lock_guard<std::mutex> g(mu);
        auto it = free_blocks.lower_bound(bytes);
        if (it != free_blocks.end() && it->first <= bytes + bytes / 2 + 4096) {
            *out = it->second;
            const size_t have = it->first;
            live_blocks[it->second] = it->first;
            free_blocks.erase(it);
            if (poison) (void)hipMemsetAsync(*out, 0xFF, have, stream);
            return SG_OK;
        }
    }
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        trim();   // give cached blocks back and retry once
        e = hipMalloc(&p, bytes);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            sg_set_error("hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
            return SG_ERR_OOM;
        }
    }
    if (poison) (void)hipMemsetAsync(p, 0xFF, bytes, stream);
    std::lock_guard<std::mutex> g(mu);
    live_blocks[p] = bytes;
    *out = p;
    return SG_OK;
}

void sg_ctx::release(void *p) {
    if (!p) return;
    std::lock_guard<std::mutex> g(mu);
    auto it = live_blocks.find(p);
    if (it == live_blocks.end()) return;
    free_blocks.emplace(it->second, p);
    live_blocks.erase(it);
}

void sg_ctx::trim() {
    std::vector<void *> to_free;
    {
        std::lock_guard<std::mutex> g(mu);
        for (auto &kv : free_blocks) to_free.push_back(kv.second);
        free_blocks.clear();
    }
    if (!to_free.empty()) (void)hipStreamSynchronize(stream);
    for (void *p : to_free) (void)hipFree(p);
}

// ------------------------------------------------------------------------------------ context
extern "C" int sg_ctx_set_option(sg_ctx *ctx, const char *name, const char *value) {
    SG_REQUIRE(ctx && name && strncmp(name, "SG_", 3) == 0, "option names start with SG_");
    std::lock_guard<std::mutex> g(ctx->mu);
    if (value) ctx->opts[name] = value;
    else ctx->opts.erase(name);
    const char *v = ctx->opt("SG_POISON_ALLOC");
    ctx->poison = v && v[0] == '1';
    return SG_OK;
}

extern "C" int sg_ctx_reset_options(sg_ctx *ctx) {
    SG_REQUIRE(ctx != nullptr, "context is null");
    std::lock_guard<std::mutex> g(ctx->mu);
    snapshot_options(ctx);
    return SG_OK;
}

extern "C" int sg_ctx_options(sg_ctx *ctx, char *buf, int64_t len) {
    if (!ctx) return 0;
    std::string all;
    {
        std::lock_guard<std::mutex> g(ctx->mu);
        for (auto &kv : ctx->opts) all += kv.first + "=" + kv.second + "\n";
    }
    if (buf && len > 0) {
        const size_t n = all.size() < (size_t)(len - 1) ? all.size() : (size_t)(len - 1);
        memcpy(buf, all.data(), n);
        buf[n] = 0;
    }
    return (int)all.size() + 1;
}

extern "C" int sg_ctx_create(int device, void *hip_stream, sg_ctx **out) {
    SG_REQUIRE(out != nullptr, "out is null");
    int n = 0;
    sg_device_count(&n);
    if (n <= 0) {
        sg_set_error("no HIP device visible: libsg_hip.so needs an MI355X (gfx950) GPU and has no CPU fallback");
        return SG_ERR_NODEVICE;
    }
    SG_REQUIRE(device >= 0 && device < n, "device index out of range");
    SG_HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    SG_HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        sg_set_error("device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
        return SG_ERR_NODEVICE;
    }
    sg_ctx *ctx = new (std::nothrow) sg_ctx();
    if (!ctx) return SG_ERR_OOM;
    snapshot_options(ctx);
    ctx->device = device;
    ctx->num_cu = prop.multiProcessorCount;
    ctx->total_mem = prop.totalGlobalMem;
    if (hip_stream) {
        ctx->stream = (hipStream_t)hip_stream;
    } else {
        SG_HIP_TRY(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
        ctx->own_stream = true;
    }
    for (int i = 0; i < SG_K_COUNT; ++i) {
        SG_HIP_TRY(hipEventCreate(&ctx->ev_start[i]));
        SG_HIP_TRY(hipEventCreate(&ctx->ev_stop[i]));
        ctx->ev_valid[i] = false;
    }
    SG_HIP_TRY(hipMalloc((void **)&ctx->d_stat_words, 8 * sizeof(int64_t)));
    SG_HIP_TRY(hipHostMalloc((void **)&ctx->h_stat_words, 8 * sizeof(int64_t), hipHostMallocDefault));
    SG_HIP_TRY(hipHostMalloc((void **)&ctx->h_fetch, SG_H_FETCH_WORDS * sizeof(uint32_t), hipHostMallocDefault));
    SG_HIP_TRY(hipMemsetAsync(ctx->d_stat_words, 0, 8 * sizeof(int64_t), ctx->stream));
    *out = ctx;
    return SG_OK;
}

extern "C" int sg_ctx_destroy(sg_ctx *ctx) {
    if (!ctx) return SG_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    ctx->trim();
    for (auto &kv : ctx->live_blocks) (void)hipFree(kv.first);
    ctx->live_blocks.clear();
    for (int i = 0; i < SG_K_COUNT; ++i) {
        (void)hipEventDestroy(ctx->ev_start[i]);
        (void)hipEventDestroy(ctx->ev_stop[i]);
    }
    (void)hipFree(ctx->d_stat_words);
    (void)hipFree(ctx->d_scan_desc);
    for (auto &t : ctx->idf_tables) (void)hipFree(t.d);
    (void)hipHostFree(ctx->h_stat_words);
    (void)hipHostFree(ctx->h_fetch);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return SG_OK;
}

extern "C" int sg_ctx_sync(sg_ctx *ctx) {
    SG_REQUIRE(ctx != nullptr, "ctx is null");
    SG_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SG_OK;
}

extern "C" int sg_ctx_trim(sg_ctx *ctx) {
    SG_REQUIRE(ctx != nullptr, "ctx is null");
    ctx->trim();
    return SG_OK;
}

extern "C" int sg_ctx_stats(sg_ctx *ctx, sg_stats *out) {
    SG_REQUIRE(ctx != nullptr && out != nullptr, "null argument");
    SG_HIP_TRY(hipMemcpyAsync(ctx->h_stat_words, ctx->d_stat_words, 8 * sizeof(int64_t), hipMemcpyDeviceToHost,
                              ctx->stream));
    SG_HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < SG_K_COUNT; ++i) {
        out->ms[i] = 0.f;
        if (ctx->ev_valid[i]) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, ctx->ev_start[i], ctx->ev_stop[i]) == hipSuccess) out->ms[i] = ms;
            else (void)hipGetLastError();
        }
    }
    out->macs = ctx->h_stat_words[0];
    out->out_nnz = ctx->h_stat_words[1];
    out->prune_rows = ctx->h_stat_words[2];
    out->prune_postings = ctx->h_stat_words[3];
    out->prune_survivors = ctx->h_stat_words[4];
    out->exact_rows = ctx->h_stat_words[5];
    out->prune_scored = ctx->h_stat_words[6];
    out->prune_bytes = out->prune_rows == 0 ? 0
                       : out->prune_postings * 4 + (int64_t)((double)out->prune_survivors * ctx->prune_q8_bytes) +
                             (int64_t)((double)out->prune_scored * ctx->prune_row_bytes) +
                             ctx->spgemm_fixed_bytes + out->out_nnz * ctx->spgemm_entry_bytes;
    out->prune_symmetric = ctx->prune_symmetric ? 1 : 0;
    out->spgemm_bytes = ctx->spgemm_fixed_bytes + (out->macs + out->out_nnz) * ctx->spgemm_entry_bytes;
    return SG_OK;
}

// ------------------------------------------------------------------------------------ strings
extern "C" int sg_strings_from_host(sg_ctx *ctx, const uint8_t *bytes, const int64_t *offsets, int64_t n,
                                    sg_strings **out) {
    SG_REQUIRE(ctx && offsets && out && n >= 0, "null argument");
    const int64_t total = offsets[n] - offsets[0];
    SG_REQUIRE(total >= 0 && (total == 0 || bytes != nullptr), "bad offsets");
    SG_REQUIRE(offsets[0] == 0, "offsets must start at 0");
    sg_strings *s = new (std::nothrow) sg_strings();
    if (!s) return SG_ERR_OOM;
    s->ctx = ctx;
    s->n = n;
    s->total_bytes = total;
    s->owned = true;
    uint8_t *db = nullptr;
    int64_t *doff = nullptr;
    int st = sg_alloc(ctx, (size_t)total + 16, &db);
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)n + 1, &doff);
    if (st != SG_OK) {
        ctx->release(db);
        delete s;
        return st;
    }
    s->d_bytes = db;
    s->d_offsets = doff;
    if (total > 0) SG_HIP_TRY(hipMemcpyAsync(db, bytes, (size_t)total, hipMemcpyHostToDevice, ctx->stream));
    SG_HIP_TRY(hipMemcpyAsync(doff, offsets, sizeof(int64_t) * (size_t)(n + 1), hipMemcpyHostToDevice, ctx->stream));
    SG_HIP_TRY(hipStreamSynchronize(ctx->stream));   // host buffers may be pageable / reused
    *out = s;
    return SG_OK;
}

extern "C" int sg_strings_from_host_symbols(sg_ctx *ctx, const uint16_t *symbols, const int64_t *offsets, int64_t n,
                                            int32_t alphabet_size, sg_strings **out) {
    SG_REQUIRE(ctx && offsets && out && n >= 0, "null argument");
    SG_REQUIRE(alphabet_size >= 1 && alphabet_size <= 65535, "alphabet_size must be in [1, 65535]");
    const int64_t total = offsets[n] - offsets[0];
    SG_REQUIRE(total >= 0 && (total == 0 || symbols != nullptr), "bad offsets");
    SG_REQUIRE(offsets[0] == 0, "offsets must start at 0");
    sg_strings *s = new (std::nothrow) sg_strings();
    if (!s) return SG_ERR_OOM;
    s->ctx = ctx;
    s->n = n;
    s->total_bytes = total;
    s->owned = true;
    s->sym_width = 2;
    s->alphabet = alphabet_size;
    s->prelowered = true;
    uint16_t *db = nullptr;
    int64_t *doff = nullptr;
    int st = sg_alloc(ctx, (size_t)total + 16, &db);
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)n + 1, &doff);
    if (st != SG_OK) {
        ctx->release(db);
        delete s;
        return st;
    }
    s->d_bytes = (const uint8_t *)db;
    s->d_offsets = doff;
    hipError_t e = hipSuccess;
    if (total > 0) e = hipMemcpyAsync(db, symbols, (size_t)total * 2, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(doff, offsets, sizeof(int64_t) * (size_t)(n + 1), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);   // host buffers may be pageable / reused
    if (e != hipSuccess) {
        sg_set_error("sg_strings_from_host_symbols: %s", hipGetErrorString(e));
        sg_strings_free(s);
        return SG_ERR_HIP;
    }
    *out = s;
    return SG_OK;
}

extern "C" int sg_strings_set_prelowered(sg_strings *s, int32_t prelowered) {
    SG_REQUIRE(s != nullptr, "strings are null");
    s->prelowered = prelowered != 0;
    return SG_OK;
}

extern "C" int sg_strings_from_device(sg_ctx *ctx, const uint8_t *d_bytes, const int64_t *d_offsets, int64_t n,
                                      int64_t total_bytes, sg_strings **out) {
    SG_REQUIRE(ctx && d_offsets && out && n >= 0 && total_bytes >= 0, "null argument");
    sg_strings *s = new (std::nothrow) sg_strings();
    if (!s) return SG_ERR_OOM;
    s->ctx = ctx;
    s->d_bytes = d_bytes;
    s->d_offsets = d_offsets;
    s->n = n;
    s->total_bytes = total_bytes;
    s->owned = false;
    *out = s;
    return SG_OK;
}

extern "C" int sg_strings_free(sg_strings *s) {
    if (!s) return SG_OK;
    if (s->owned) {
        s->ctx->release((void *)s->d_bytes);
        s->ctx->release((void *)s->d_offsets);
    }
    delete s;
    return SG_OK;
}

// ------------------------------------------------------------------------------------ CSR
static size_t dtype_size(int32_t dtype) { return dtype == SG_F64 ? 8 : 4; }

extern "C" int sg_csr_from_host(sg_ctx *ctx, int64_t n_rows, int64_t n_cols, const int64_t *indptr,
                                const int32_t *indices, const void *data, int32_t dtype, sg_csr **out) {
    SG_REQUIRE(ctx && indptr && out, "null argument");
    SG_REQUIRE(n_rows >= 0 && n_cols >= 0, "negative shape");
    SG_REQUIRE(dtype == SG_F32 || dtype == SG_F64, "dtype must be SG_F32 or SG_F64");
    SG_REQUIRE(indptr[0] == 0, "indptr must start at 0");
    const int64_t nnz = indptr[n_rows];
    SG_REQUIRE(nnz >= 0 && (nnz == 0 || (indices && data)), "bad nnz");
    if (n_cols > INT32_MAX || n_rows > INT32_MAX) {
        sg_set_error("matrix shape (%lld x %lld) exceeds int32 indices", (long long)n_rows, (long long)n_cols);
        return SG_ERR_OVERFLOW;
    }
    sg_csr *m = new (std::nothrow) sg_csr();
    if (!m) return SG_ERR_OOM;
    m->ctx = ctx;
    m->n_rows = n_rows;
    m->n_cols = n_cols;
    m->nnz = nnz;
    m->dtype = dtype;
    m->owned = true;
    int64_t *dp = nullptr;
    int32_t *di = nullptr;
    void *dd = nullptr;
    int st = sg_alloc(ctx, (size_t)n_rows + 1, &dp);
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)nnz + 4, &di);
    if (st == SG_OK) st = ctx->alloc(((size_t)nnz + 4) * dtype_size(dtype), &dd);
    if (st != SG_OK) {
        ctx->release(dp);
        ctx->release(di);
        delete m;
        return st;
    }
    m->d_indptr = dp;
    m->d_indices = di;
    m->d_data = dd;
    SG_HIP_TRY(hipMemcpyAsync(dp, indptr, sizeof(int64_t) * (size_t)(n_rows + 1), hipMemcpyHostToDevice, ctx->stream));
    if (nnz > 0) {
        SG_HIP_TRY(hipMemcpyAsync(di, indices, sizeof(int32_t) * (size_t)nnz, hipMemcpyHostToDevice, ctx->stream));
        SG_HIP_TRY(hipMemcpyAsync(dd, data, dtype_size(dtype) * (size_t)nnz, hipMemcpyHostToDevice, ctx->stream));
    }
    SG_HIP_TRY(hipStreamSynchronize(ctx->stream));
    *out = m;
    return SG_OK;
}

extern "C" int sg_csr_from_device(sg_ctx *ctx, int64_t n_rows, int64_t n_cols, int64_t nnz, const int64_t *d_indptr,
                                  const int32_t *d_indices, const void *d_data, int32_t dtype, sg_csr **out) {
    SG_REQUIRE(ctx && d_indptr && out, "null argument");
    SG_REQUIRE(n_rows >= 0 && n_cols >= 0 && nnz >= 0, "negative shape");
    SG_REQUIRE(dtype == SG_F32 || dtype == SG_F64, "dtype must be SG_F32 or SG_F64");
    if (n_cols > INT32_MAX || n_rows > INT32_MAX) {
        sg_set_error("matrix shape exceeds int32 indices");
        return SG_ERR_OVERFLOW;
    }
    sg_csr *m = new (std::nothrow) sg_csr();
    if (!m) return SG_ERR_OOM;
    m->ctx = ctx;
    m->n_rows = n_rows;
    m->n_cols = n_cols;
    m->nnz = nnz;
    m->dtype = dtype;
    m->d_indptr = d_indptr;
    m->d_indices = d_indices;
    m->d_data = d_data;
    m->owned = false;
    *out = m;
    return SG_OK;
}

extern "C" int sg_csr_dims(const sg_csr *m, int64_t *n_rows, int64_t *n_cols, int64_t *nnz, int32_t *dtype) {
    SG_REQUIRE(m != nullptr, "matrix is null");
    if (n_rows) *n_rows = m->n_rows;
    if (n_cols) *n_cols = m->n_cols;
    if (nnz) *nnz = m->nnz;
    if (dtype) *dtype = m->dtype;
    return SG_OK;
}

extern "C" int sg_csr_device_ptrs(const sg_csr *m, const int64_t **d_indptr, const int32_t **d_indices,
                                  const void **d_data) {
    SG_REQUIRE(m != nullptr, "matrix is null");
    if (d_indptr) *d_indptr = m->d_indptr;
    if (d_indices) *d_indices = m->d_indices;
    if (d_data) *d_data = m->d_data;
    return SG_OK;
}

extern "C" int sg_csr_to_host(sg_ctx *ctx, const sg_csr *m, int64_t *indptr, int32_t *indices, void *data) {
    SG_REQUIRE(ctx && m && indptr, "null argument");
    SG_HIP_TRY(hipMemcpyAsync(indptr, m->d_indptr, sizeof(int64_t) * (size_t)(m->n_rows + 1), hipMemcpyDeviceToHost,
                              ctx->stream));
    SG_HIP_TRY(hipStreamSynchronize(ctx->stream));
    // a row-block view has absolute offsets: copy [indptr[0], indptr[n]) and rebase
    const int64_t base = indptr[0];
    const int64_t nnz = indptr[m->n_rows] - base;
    if (nnz > 0) {
        SG_REQUIRE(indices && data, "null output");
        SG_HIP_TRY(hipMemcpyAsync(indices, m->d_indices + base, sizeof(int32_t) * (size_t)nnz, hipMemcpyDeviceToHost,
                                  ctx->stream));
        SG_HIP_TRY(hipMemcpyAsync(data, (const char *)m->d_data + dtype_size(m->dtype) * (size_t)base,
                                  dtype_size(m->dtype) * (size_t)nnz, hipMemcpyDeviceToHost, ctx->stream));
        SG_HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    if (base != 0)
        for (int64_t i = 0; i <= m->n_rows; ++i) indptr[i] -= base;
    return SG_OK;
}

extern "C" int sg_csr_row_block(sg_ctx *ctx, const sg_csr *m, int64_t r0, int64_t r1, sg_csr **out) {
    SG_REQUIRE(ctx && m && out, "null argument");
    SG_REQUIRE(0 <= r0 && r0 <= r1 && r1 <= m->n_rows, "row range out of bounds");
    sg_csr *v = new (std::nothrow) sg_csr();
    if (!v) return SG_ERR_OOM;
    *v = *m;
    v->owned = false;
    // a view owns its own groups of identical rows (sg_spgemm_topn makes them on first use): the parent's are indexed by
    // the PARENT's rows, and sg_csr_free of the view would free them under the parent
    v->left_groups = nullptr;
    v->left_state = 0;
    v->n_rows = r1 - r0;
    v->d_indptr = m->d_indptr + r0;
    int64_t ends[2] = {0, 0};
    SG_HIP_TRY(hipMemcpyAsync(&ends[0], m->d_indptr + r0, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    SG_HIP_TRY(hipMemcpyAsync(&ends[1], m->d_indptr + r1, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    SG_HIP_TRY(hipStreamSynchronize(ctx->stream));
    v->nnz = ends[1] - ends[0];
    *out = v;
    return SG_OK;
}

extern "C" int sg_csr_free(sg_csr *m) {
    if (!m) return SG_OK;
    if (m->left_groups) sg_collapse_free(m->left_groups);
    if (m->owned) {
        m->ctx->release((void *)m->d_indptr);
        m->ctx->release((void *)m->d_indices);
        m->ctx->release((void *)m->d_data);
        m->ctx->release(m->d_props_words);
    }
    delete m;
    return SG_OK;
}

// ------------------------------------------------------------------------------------ zeroing several arrays at once
// Counters and per-row count arrays are cleared in groups at the head of a call; every hipMemsetAsync is a launch of its
// own (a fillBuffer kernel: 33 of the step's 135 launches at 663 k).  One kernel clears up to eight ranges (4-byte
// granularity, 16-byte stores where the range is aligned).
struct SgZeroJobs {
    uint32_t *ptr[8];
    uint64_t words[8];    // 4-byte words of range i
    uint64_t first[8];    // words of the ranges before i (prefix sum); first[n] = all
    int n;
};
__global__ void __launch_bounds__(256) zero_ranges_kernel(SgZeroJobs jobs) {
    const uint64_t total = jobs.first[jobs.n < 8 ? jobs.n : 7] + (jobs.n == 8 ? jobs.words[7] : 0);
    // a thread clears four consecutive words of one range (the ranges are padded to multiples of four in this numbering)
    for (uint64_t q = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; q < total; q += (uint64_t)gridDim.x * blockDim.x * 4) {
        int r = 0;
#pragma unroll
        for (int i = 1; i < 8; ++i)
            if (i < jobs.n && q >= jobs.first[i]) r = i;
        const uint64_t w = q - jobs.first[r];
        uint32_t *p = jobs.ptr[r] + w;
        const uint64_t left = jobs.words[r] - w;
        if (left >= 4 && ((uintptr_t)p & 15u) == 0) {
            *reinterpret_cast<uint4 *>(p) = make_uint4(0u, 0u, 0u, 0u);
        } else {
            for (uint64_t e = 0; e < 4 && e < left; ++e) p[e] = 0u;
        }
    }
}

int sg_zero_ranges(sg_ctx *ctx, int n, void *const *ptrs, const size_t *bytes) {
    SgZeroJobs jobs;
    jobs.n = 0;
    uint64_t first = 0;
    for (int i = 0; i < n; ++i) {
        if (!ptrs[i] || bytes[i] == 0) continue;
        if (jobs.n == 8 || (bytes[i] & 3u) || ((uintptr_t)ptrs[i] & 3u)) {   // (not expected: plain memsets serve)
            SG_HIP_TRY(hipMemsetAsync(ptrs[i], 0, bytes[i], ctx->stream));
            continue;
        }
        jobs.ptr[jobs.n] = (uint32_t *)ptrs[i];
        jobs.words[jobs.n] = bytes[i] / 4;
        jobs.first[jobs.n] = first;
        first += (jobs.words[jobs.n] + 3) & ~(uint64_t)3;
        ++jobs.n;
    }
    if (jobs.n == 0) return SG_OK;
    for (int i = jobs.n; i < 8; ++i) {
        jobs.ptr[i] = nullptr;
        jobs.words[i] = 0;
        jobs.first[i] = first;
    }
    uint64_t threads = first / 4;
    unsigned grid = (unsigned)((threads + 255) / 256);
    if (grid > (unsigned)ctx->num_cu * 16u) grid = (unsigned)ctx->num_cu * 16u;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(zero_ranges_kernel, dim3(grid), dim3(256), 0, ctx->stream, jobs);
    SG_HIP_TRY(hipGetLastError());
    return SG_OK;
}

// ------------------------------------------------------------------------------------ prefix sums (sg_scan.h)
int sg_exclusive_scan_u32(sg_ctx *ctx, const uint32_t *d_in, uint32_t *d_out, int64_t n, uint32_t *d_total) {
    return scan_impl<uint32_t, uint32_t>(ctx, d_in, d_out, n, d_total);
}

int sg_exclusive_scan_positive_i32(sg_ctx *ctx, const int32_t *d_in, uint32_t *d_out, int64_t n, uint32_t *d_total) {
    return scan_impl<int32_t, uint32_t, true>(ctx, d_in, d_out, n, d_total);
}

int sg_exclusive_scan_i32_to_i64(sg_ctx *ctx, const int32_t *d_in, int64_t *d_out, int64_t n) {
    // d_out has n + 1 entries; d_out[n] = total
    return scan_impl<int32_t, int64_t>(ctx, d_in, d_out, n, d_out + n);
}
