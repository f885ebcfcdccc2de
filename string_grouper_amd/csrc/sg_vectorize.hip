// K1 -- character n-gram tokeniser + vocabulary / document-frequency build
// K2 -- tf * idf weighting + row L2 normalisation, emitting the CSR TF-IDF matrix
//
// Replaces, for the reference's TfidfVectorizer(min_df=1, analyzer=self.n_grams, dtype=...)
// (string_grouper/string_grouper.py:306, :365-378, :685-707):
//   StringGrouper.n_grams                    string_grouper.py:365-378   (ASCII path; lower + regex delete)
//   CountVectorizer._count_vocab/_sort_features  sklearn feature_extraction/text.py:1247-1310, :1194-1206
//   TfidfTransformer.fit / transform          text.py:1636-1681, :1683-1724
//   _inplace_csr_row_normalize_l2             sklearn utils/sparsefuncs_fast.pyx:572-598
//
// Key idea.  After the host's lower()/NFKD step every surviving character is 7-bit ASCII, so an
// n-gram is a fixed-length byte string and its big-endian packing (bits_per_char bits per character)
// is an integer whose numeric order equals sklearn's sorted-vocabulary (code point) order.  No string
// vocabulary is ever materialised: column id = rank of the key among the keys that occur.
//
// K1: one 64-lane wave per string.  Bytes are filtered (>= 0x80 dropped == .encode('ascii','ignore');
// regex character class == 128-entry delete table; optional ASCII lower) and compacted with
// ballot + prefix popcount into LDS, each lane packs the n-grams starting at its positions, the wave
// sorts the keys with a bitonic network in LDS and run-length encodes them: (key, tf) ascending ==
// exactly the row sklearn's _count_vocab + sort_indices produces.  Rows are written into a padded
// layout whose offsets come from the string lengths alone (a row of L bytes has at most L-n+1
// n-grams), so tokenising is a single pass with coalesced writes.
// df: one global atomic per (row, distinct key) into a dense table over the key space; vocabulary:
// exclusive scan over (df > 0).
// K2: a thread per row: col = rank[key], w = (T)tf * idf[col] (one rounding), acc(double) += (T)(w*w)
// sequentially in column order, w = (T)((double)w / sqrt(acc)); rows with acc == 0 are left alone.
// This op order reproduces sklearn bit for bit (tests/test_parity_gpu.py).
//
// Bound: HBM.  Algorithmic bytes = sum(len) + 8n (read strings) + nnz*(4+s) + 8(n+1) (write CSR)
// + the key-space tables (4 * 2^(bits*n) for df and rank).
#include "sg_internal.h"

#define TOK_CAP 1024            // n-grams per string the device tokeniser handles
#define SG_KEY_OOV 0xFFFFFFFEu   // key of an n-gram that contains such a character (sorts behind every real key)
#define SG_CHAR_ABSENT 0xFF     // character code of a byte that did not occur at fit(): its n-grams are out of vocabulary
#define TOK_CHARS (TOK_CAP + 16)

struct TokenCache {             // tokenised strings in the padded layout
    const sg_strings *src = nullptr;
    int64_t n = 0;
    int64_t cap_total = 0;
    int64_t *d_ub_ptr = nullptr;   // n + 1: start of row i's slots
    int32_t *d_cnt = nullptr;      // n: distinct n-grams of row i
    uint32_t *d_keys = nullptr;    // cap_total
    int32_t *d_tf = nullptr;       // cap_total
};

struct VocabImpl {
    std::vector<TokenCache> caches;
    uint8_t rank_of_byte[128];     // byte -> compact character code
    uint8_t byte_of_rank[128];
    int32_t *d_df_table = nullptr; // key_space
    int32_t *d_err = nullptr;      // [0] != 0: a string exceeded TOK_CAP
    bool local_alphabet = false;   // characters coded by rank among the bytes seen at fit() (ngram_size > 3)
};

static VocabImpl *impl_of(const sg_vocab *v) { return v ? v->impl : nullptr; }

struct TokParams {
    int32_t ngram;
    int32_t ascii_lower;
    int32_t bits;
    uint32_t del_mask[4];          // bit c set: ASCII byte c is deleted
    uint8_t rank_of_byte[128];
};

// -------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) ub_count_kernel(const int64_t *__restrict__ offsets, int64_t n, int32_t ngram,
                                                       int32_t *ub) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t len = offsets[i + 1] - offsets[i];
    int64_t g = len - ngram + 1;
    if (g < 0) g = 0;
    if (g > TOK_CAP) g = TOK_CAP;   // longer rows are reported through the error word by the tokeniser
    ub[i] = (int32_t)g;
}

// alphabet presence: which filtered byte values occur at all (only needed when 7*n bits is too wide)
__global__ void __launch_bounds__(256) alphabet_kernel(const uint8_t *__restrict__ bytes, int64_t total, TokParams p,
                                                       uint32_t *present /*[4]*/) {
    __shared__ uint32_t local[4];
    if (threadIdx.x < 4) local[threadIdx.x] = 0;
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t c = bytes[i];
        if (c >= 0x80) continue;
        if (p.ascii_lower && c >= 'A' && c <= 'Z') c += 32;
        if ((p.del_mask[c >> 5] >> (c & 31)) & 1u) continue;
        atomicOr(&local[c >> 5], 1u << (c & 31));
    }
    __syncthreads();
    if (threadIdx.x < 4 && local[threadIdx.x]) atomicOr(&present[threadIdx.x], local[threadIdx.x]);
}

// One wave per string: filter, n-gram, sort, run-length encode.
// The tokeniser runs single-wave workgroups: LDS operations of one wave execute in issue order, so its
// phases need no s_barrier (and no wait for the outstanding df atomics / token stores of the previous
// string, which __syncthreads() implies); a compiler-level barrier keeps the LDS accesses in program order.
__device__ __forceinline__ void wave_sync() { __builtin_amdgcn_wave_barrier(); }

__global__ void __launch_bounds__(64) tokenize_kernel(const uint8_t *__restrict__ bytes,
                                                      const int64_t *__restrict__ offsets, int64_t n_rows,
                                                      TokParams p, const int64_t *__restrict__ ub_ptr,
                                                      int32_t *__restrict__ out_cnt, uint32_t *__restrict__ out_keys,
                                                      int32_t *__restrict__ out_tf, int32_t *df_table,
                                                      int32_t df_replicas, int64_t df_stride, int32_t *err) {
    __shared__ uint32_t keys[TOK_CAP];
    __shared__ uint16_t starts[TOK_CAP + 2];
    __shared__ uint8_t chars[TOK_CHARS];
    const int lane = threadIdx.x;
    const uint64_t lt_mask = ((uint64_t)1 << lane) - 1;

    for (int64_t row = blockIdx.x; row < n_rows; row += gridDim.x) {
        const int64_t b0 = offsets[row];
        const int64_t len = offsets[row + 1] - b0;
        // ---- filter + compact into LDS
        int m = 0;   // surviving characters
        bool overflow = false;
        for (int64_t base = 0; base < len; base += 64) {
            uint32_t c = 0xFF;
            if (base + lane < len) c = bytes[b0 + base + lane];
            if (p.ascii_lower && c >= 'A' && c <= 'Z') c += 32;
            const bool keep = c < 0x80 && !((p.del_mask[(c & 127) >> 5] >> (c & 31)) & 1u);
            const uint64_t km = __ballot(keep);
            const int pos = m + __popcll(km & lt_mask);
            if (keep && pos < TOK_CHARS) chars[pos] = p.rank_of_byte[c];
            m += __popcll(km);
            if (m > TOK_CHARS) {
                overflow = true;
                break;
            }
        }
        wave_sync();
        int g = m - p.ngram + 1;   // number of n-grams
        if (g < 0) g = 0;
        if (overflow || g > TOK_CAP) {
            if (lane == 0) atomicExch(err, 1);
            if (lane == 0) out_cnt[row] = 0;
            wave_sync();
            continue;
        }
        if (g == 0) {
            if (lane == 0) out_cnt[row] = 0;
            wave_sync();
            continue;
        }
        // ---- pack keys
        int P = 64;
        while (P < g) P <<= 1;
        for (int idx = lane; idx < P; idx += 64) {
            uint32_t key = 0xFFFFFFFFu;
            if (idx < g) {
                key = 0;
                bool absent = false;   // a character that did not occur at fit(): the n-gram is out of vocabulary
                for (int q = 0; q < p.ngram; ++q) {
                    const uint32_t ch = chars[idx + q];
                    absent |= ch == SG_CHAR_ABSENT;
                    key = (key << p.bits) | ch;
                }
                if (absent) key = SG_KEY_OOV;
            }
            keys[idx] = key;
        }
        wave_sync();
        // ---- bitonic sort, ascending
        for (int k = 2; k <= P; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int idx = lane; idx < (P >> 1); idx += 64) {
                    const int i = ((idx & ~(j - 1)) << 1) | (idx & (j - 1));
                    const uint32_t a = keys[i], b = keys[i + j];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) {
                        keys[i] = b;
                        keys[i + j] = a;
                    }
                }
                wave_sync();
            }
        }
        // ---- run-length encode: starts[u] = index of the first occurrence of the u-th distinct key
        int uniq = 0;
        for (int base = 0; base < g; base += 64) {
            const int idx = base + lane;
            bool head = false;
            if (idx < g) head = (idx == 0) || (keys[idx] != keys[idx - 1]);
            const uint64_t hm = __ballot(head);
            if (head) starts[uniq + __popcll(hm & lt_mask)] = (uint16_t)idx;
            uniq += __popcll(hm);
        }
        if (lane == 0) starts[uniq] = (uint16_t)g;
        wave_sync();
        const int64_t obase = ub_ptr[row];
        for (int u = lane; u < uniq; u += 64) {
            const int s0 = starts[u];
            const uint32_t key = keys[s0];
            out_keys[obase + u] = key;
            out_tf[obase + u] = (int32_t)starts[u + 1] - s0;
            // the same few n-grams ('inc', ' co') occur in a fifth of all strings: spread their atomics over
            // several copies of the table (summed by df_reduce_kernel) instead of serialising on one address
            if (df_table && key != SG_KEY_OOV) atomicAdd(&df_table[(int64_t)(row % df_replicas) * df_stride + key], 1);
        }
        if (lane == 0) out_cnt[row] = uniq;
        wave_sync();
    }
}

__global__ void __launch_bounds__(256) df_reduce_kernel(int32_t *df_table, int64_t key_space, int32_t replicas, int64_t stride) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= key_space) return;
    int32_t s = df_table[i];
    for (int r = 1; r < replicas; ++r) s += df_table[(int64_t)r * stride + i];
    df_table[i] = s;
}

__global__ void __launch_bounds__(256) presence_kernel(const int32_t *__restrict__ df_table, int64_t key_space,
                                                       uint32_t *present) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < key_space) present[i] = df_table[i] > 0 ? 1u : 0u;
}

// rank[key] = column id (exclusive scan of presence) or -1; also the column -> key / df arrays
__global__ void __launch_bounds__(256) vocab_finalize_kernel(const int32_t *__restrict__ df_table, int64_t key_space,
                                                             int32_t *rank_io /* in: exclusive scan, out: col or -1 */,
                                                             uint64_t *col_keys, int32_t *col_df) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= key_space) return;
    const int32_t df = df_table[i];
    const int32_t r = rank_io[i];
    if (df > 0) {
        col_keys[r] = (uint64_t)i;
        col_df[r] = df;
    } else {
        rank_io[i] = -1;
    }
}

__global__ void __launch_bounds__(256) kept_count_kernel(const int64_t *__restrict__ ub_ptr,
                                                         const int32_t *__restrict__ cnt,
                                                         const uint32_t *__restrict__ keys,
                                                         const int32_t *__restrict__ key_to_col, int64_t n,
                                                         int32_t *kept) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t b = ub_ptr[i];
    const int c = cnt[i];
    int k = 0;
    for (int q = 0; q < c; ++q) {
        const uint32_t key = keys[b + q];
        k += key != SG_KEY_OOV && key_to_col[key] >= 0;
    }
    kept[i] = k;
}

template <typename T>
__device__ __forceinline__ T tmul(T a, T b);
template <>
__device__ __forceinline__ float tmul<float>(float a, float b) { return __fmul_rn(a, b); }
template <>
__device__ __forceinline__ double tmul<double>(double a, double b) { return __dmul_rn(a, b); }

template <typename T>
__global__ void __launch_bounds__(256) weight_normalize_kernel(const int64_t *__restrict__ ub_ptr,
                                                               const int32_t *__restrict__ cnt,
                                                               const uint32_t *__restrict__ keys,
                                                               const int32_t *__restrict__ tf,
                                                               const int32_t *__restrict__ key_to_col,
                                                               const T *__restrict__ idf, int64_t n,
                                                               const int64_t *__restrict__ indptr,
                                                               int32_t *__restrict__ out_idx, T *__restrict__ out_val) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t b = ub_ptr[i];
    const int c = cnt[i];
    int64_t o = indptr[i];
    const int64_t o0 = o;
    double acc = 0.0;
    for (int q = 0; q < c; ++q) {   // ascending key == ascending column
        const uint32_t key = keys[b + q];
        const int32_t col = key != SG_KEY_OOV ? key_to_col[key] : -1;
        if (col < 0) continue;       // out-of-vocabulary n-gram of a string that was not part of fit()
        const T w = tmul<T>((T)tf[b + q], idf[col]);
        out_idx[o] = col;
        out_val[o] = w;
        acc = __dadd_rn(acc, (double)tmul<T>(w, w));
        ++o;
    }
    if (acc == 0.0) return;
    const double nrm = __dsqrt_rn(acc);
    for (int64_t q = o0; q < o; ++q) out_val[q] = (T)__ddiv_rn((double)out_val[q], nrm);
}

// -------------------------------------------------------------------------------------------------
static void free_cache(sg_ctx *ctx, TokenCache &c) {
    ctx->release(c.d_ub_ptr);
    ctx->release(c.d_cnt);
    ctx->release(c.d_keys);
    ctx->release(c.d_tf);
    c = TokenCache();
}

static int tokenize_set(sg_ctx *ctx, const sg_strings *s, const TokParams &tp, int32_t *df_table, int32_t df_replicas,
                        int64_t df_stride, int32_t *d_err,
                        TokenCache *out) {
    TokenCache c;
    c.src = s;
    c.n = s->n;
    int32_t *ub = nullptr;
    int st = sg_alloc(ctx, (size_t)s->n + 1, &ub);
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)s->n + 1, &c.d_ub_ptr);
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)s->n + 1, &c.d_cnt);
    if (st == SG_OK && s->n > 0) {
        hipLaunchKernelGGL(ub_count_kernel, dim3((unsigned)((s->n + 255) / 256)), dim3(256), 0, ctx->stream,
                           s->d_offsets, s->n, tp.ngram, ub);
        st = sg_exclusive_scan_i32_to_i64(ctx, ub, c.d_ub_ptr, s->n);
    } else if (st == SG_OK) {
        if (hipMemsetAsync(c.d_ub_ptr, 0, sizeof(int64_t), ctx->stream) != hipSuccess) st = SG_ERR_HIP;
    }
    ctx->release(ub);
    // every row of L bytes has at most L - n + 1 n-grams, so total_bytes bounds the padded size
    c.cap_total = s->total_bytes + 1;
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)c.cap_total, &c.d_keys);
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)c.cap_total, &c.d_tf);
    if (st == SG_OK && s->n > 0) {
        unsigned grid = (unsigned)ctx->num_cu * 24u;
        if ((int64_t)grid > s->n) grid = (unsigned)s->n;
        hipLaunchKernelGGL(tokenize_kernel, dim3(grid), dim3(64), 0, ctx->stream, s->d_bytes, s->d_offsets, s->n, tp,
                           (const int64_t *)c.d_ub_ptr, c.d_cnt, c.d_keys, c.d_tf, df_table, df_replicas, df_stride, d_err);
        if (hipGetLastError() != hipSuccess) {
            sg_set_error("tokenize_kernel launch failed");
            st = SG_ERR_HIP;
        }
    }
    if (st != SG_OK) {
        free_cache(ctx, c);
        return st;
    }
    *out = c;
    return SG_OK;
}

static TokParams make_tok_params(const sg_vocab *v, const VocabImpl *im) {
    TokParams tp;
    memset(&tp, 0, sizeof(tp));
    tp.ngram = v->params.ngram_size;
    tp.ascii_lower = v->params.ascii_lower;
    tp.bits = v->bits_per_char;
    for (int c = 0; c < 128; ++c)
        if (v->params.delete_table[c]) tp.del_mask[c >> 5] |= 1u << (c & 31);
    memcpy(tp.rank_of_byte, im->rank_of_byte, 128);
    return tp;
}

// fit = begin (tokenise, count document frequencies into the dense key table) + end (vocabulary from the table).
// The two halves are separate entry points so that a multi-GPU caller can all-reduce the table in between
// (one rank tokenises one block of the strings; every rank then derives the SAME vocabulary and idf).
extern "C" int sg_vec_fit_begin(sg_ctx *ctx, const sg_strings *const *sets, int32_t n_sets, const sg_vec_params *params,
                                sg_vocab **out) {
    SG_REQUIRE(ctx && sets && params && out && n_sets >= 1, "null argument");
    SG_REQUIRE(params->ngram_size >= 1 && params->ngram_size <= 9, "ngram_size must be in [1, 9]");
    SG_REQUIRE(params->dtype == SG_F32 || params->dtype == SG_F64, "dtype must be SG_F32 or SG_F64");
    for (int i = 0; i < n_sets; ++i) SG_REQUIRE(sets[i] != nullptr, "null string set");

    sg_vocab *v = new (std::nothrow) sg_vocab();
    VocabImpl *im = new (std::nothrow) VocabImpl();
    if (!v || !im) {
        delete v;
        delete im;
        return SG_ERR_OOM;
    }
    v->ctx = ctx;
    v->params = *params;
    v->impl = im;
    int st = SG_OK;
    // ---- character coding: raw 7 bits when the key space stays small, else ranks of the bytes present
    for (int c = 0; c < 128; ++c) {
        im->rank_of_byte[c] = (uint8_t)c;
        im->byte_of_rank[c] = (uint8_t)c;
    }
    v->bits_per_char = 7;
    TokParams tp = make_tok_params(v, im);
    if (7 * params->ngram_size > 24) {
        uint32_t *d_present = nullptr;
        st = sg_alloc(ctx, 4, &d_present);
        uint32_t present[4] = {0, 0, 0, 0};
        if (st == SG_OK) {
            (void)hipMemsetAsync(d_present, 0, 16, ctx->stream);
            for (int i = 0; i < n_sets; ++i)
                if (sets[i]->total_bytes > 0)
                    hipLaunchKernelGGL(alphabet_kernel, dim3(1024), dim3(256), 0, ctx->stream, sets[i]->d_bytes,
                                       sets[i]->total_bytes, tp, d_present);
            if (hipMemcpyAsync(present, d_present, 16, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                hipStreamSynchronize(ctx->stream) != hipSuccess)
                st = SG_ERR_HIP;
        }
        ctx->release(d_present);
        if (st == SG_OK) {
            // bytes that do not occur get the out-of-alphabet code: an n-gram containing one (only possible in a
            // later transform() of strings that were not part of fit()) is not in the vocabulary and is skipped
            for (int c = 0; c < 128; ++c) im->rank_of_byte[c] = SG_CHAR_ABSENT;
            int sigma = 0;
            for (int c = 0; c < 128; ++c)
                if ((present[c >> 5] >> (c & 31)) & 1u) {
                    im->rank_of_byte[c] = (uint8_t)sigma;
                    im->byte_of_rank[sigma] = (uint8_t)c;
                    ++sigma;
                }
            int bits = 1;
            while ((1 << bits) < sigma) ++bits;
            v->bits_per_char = bits;
            im->local_alphabet = true;
            if (bits * params->ngram_size > 30) {
                sg_set_error("n-gram key space 2^%d (alphabet of %d characters, ngram_size %d) is too large for the "
                             "device vocabulary table", bits * params->ngram_size, sigma, params->ngram_size);
                st = SG_ERR_UNSUPPORTED;
            }
            tp = make_tok_params(v, im);
        }
    }
    if (st != SG_OK) {
        sg_vocab_free(v);
        return st;
    }
    v->key_space = (int64_t)1 << (v->bits_per_char * params->ngram_size);

    {
        SgTimer timer(ctx, SG_K_TOKENIZE);
        // df table: `replicas` copies (row mod replicas picks one) while they stay small, summed afterwards
        const int64_t df_stride = v->key_space + 1;
        int32_t replicas = 8;
        if (const char *e = getenv("SG_DF_REPLICAS")) replicas = atoi(e);
        while (replicas > 1 && df_stride * replicas > ((int64_t)1 << 25)) replicas >>= 1;   // <= 128 MiB of counters
        if (replicas < 1) replicas = 1;
        st = sg_alloc(ctx, (size_t)(df_stride * replicas), &im->d_df_table);
        if (st == SG_OK) st = sg_alloc(ctx, (size_t)v->key_space + 1, &v->d_key_to_col);
        if (st == SG_OK) st = sg_alloc(ctx, 4, &im->d_err);
        if (st == SG_OK) {
            (void)hipMemsetAsync(im->d_df_table, 0, sizeof(int32_t) * (size_t)(df_stride * replicas), ctx->stream);
            (void)hipMemsetAsync(im->d_err, 0, 16, ctx->stream);
        }
        for (int i = 0; i < n_sets && st == SG_OK; ++i) {
            TokenCache c;
            st = tokenize_set(ctx, sets[i], tp, im->d_df_table, replicas, df_stride, im->d_err, &c);
            if (st == SG_OK) {
                im->caches.push_back(c);
                v->n_docs += sets[i]->n;
            }
        }
        if (st == SG_OK && replicas > 1) {
            const unsigned grid = (unsigned)((v->key_space + 255) / 256);
            hipLaunchKernelGGL(df_reduce_kernel, dim3(grid), dim3(256), 0, ctx->stream, im->d_df_table, v->key_space,
                               replicas, df_stride);
            if (hipGetLastError() != hipSuccess) st = SG_ERR_HIP;
        }
    }
    if (st != SG_OK) {
        sg_vocab_free(v);
        return st;
    }
    *out = v;
    return SG_OK;
}

extern "C" int sg_vocab_df_table(sg_vocab *v, int32_t **d_table, int64_t *n_entries, int32_t *shareable) {
    SG_REQUIRE(v && v->impl && d_table && n_entries, "null argument");
    SG_REQUIRE(v->n_terms == 0, "the vocabulary is already finished");
    *d_table = v->impl->d_df_table;
    *n_entries = v->key_space;
    // a table coded with the alphabet of the LOCAL strings (ngram_size > 3) means something else on every rank
    if (shareable) *shareable = v->impl->local_alphabet ? 0 : 1;
    return SG_OK;
}

extern "C" int sg_vec_fit_end(sg_ctx *ctx, sg_vocab *v, int64_t n_docs_total) {
    SG_REQUIRE(ctx && v && v->impl, "null argument");
    SG_REQUIRE(v->n_terms == 0, "the vocabulary is already finished");
    VocabImpl *im = v->impl;
    if (n_docs_total > 0) v->n_docs = n_docs_total;
    int st = SG_OK;
    {
        SgTimer timer(ctx, SG_K_VOCAB);
        // ---- vocabulary = keys with df > 0, column id = rank
        uint32_t *d_total = nullptr;
        st = sg_alloc(ctx, 4, &d_total);
        if (st == SG_OK) {
            const unsigned grid = (unsigned)((v->key_space + 255) / 256);
            hipLaunchKernelGGL(presence_kernel, dim3(grid), dim3(256), 0, ctx->stream, im->d_df_table, v->key_space,
                               (uint32_t *)v->d_key_to_col);
            st = sg_exclusive_scan_u32(ctx, (const uint32_t *)v->d_key_to_col, (uint32_t *)v->d_key_to_col,
                                       v->key_space, d_total);
        }
        uint32_t host_words[2] = {0, 0};
        if (st == SG_OK) {
            if (hipMemcpyAsync(&host_words[0], d_total, 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                hipMemcpyAsync(&host_words[1], im->d_err, 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                hipStreamSynchronize(ctx->stream) != hipSuccess) {
                sg_set_error("reading the vocabulary size failed: %s", hipGetErrorString(hipGetLastError()));
                st = SG_ERR_HIP;
            }
        }
        ctx->release(d_total);
        if (st == SG_OK && host_words[1] != 0) {
            sg_set_error("a string has more than %d n-grams; the device tokeniser does not handle it", TOK_CAP);
            st = SG_ERR_UNSUPPORTED;
        }
        if (st == SG_OK) {
            v->n_terms = host_words[0];
            if (v->n_terms == 0) {
                sg_set_error("empty vocabulary; perhaps the documents only contain stop words");
                st = SG_ERR_BADARG;
            }
        }
        if (st == SG_OK) st = sg_alloc(ctx, (size_t)v->n_terms + 1, &v->d_keys);
        if (st == SG_OK) st = sg_alloc(ctx, (size_t)v->n_terms + 1, &v->d_df);
        if (st == SG_OK) {
            const unsigned grid = (unsigned)((v->key_space + 255) / 256);
            hipLaunchKernelGGL(vocab_finalize_kernel, dim3(grid), dim3(256), 0, ctx->stream, im->d_df_table,
                               v->key_space, v->d_key_to_col, v->d_keys, v->d_df);
            if (hipGetLastError() != hipSuccess) st = SG_ERR_HIP;
        }
    }
    return st;
}

extern "C" int sg_vec_fit(sg_ctx *ctx, const sg_strings *const *sets, int32_t n_sets, const sg_vec_params *params,
                          sg_vocab **out) {
    SG_REQUIRE(out != nullptr, "null argument");
    sg_vocab *v = nullptr;
    SG_TRY(sg_vec_fit_begin(ctx, sets, n_sets, params, &v));
    const int st = sg_vec_fit_end(ctx, v, 0);
    if (st != SG_OK) {
        sg_vocab_free(v);
        return st;
    }
    *out = v;
    return SG_OK;
}

extern "C" int sg_vocab_size(const sg_vocab *v, int64_t *n_terms, int64_t *n_docs) {
    SG_REQUIRE(v != nullptr, "vocab is null");
    if (n_terms) *n_terms = v->n_terms;
    if (n_docs) *n_docs = v->n_docs;
    return SG_OK;
}

extern "C" int sg_vocab_to_host(sg_ctx *ctx, const sg_vocab *v, uint64_t *keys, int64_t *df) {
    SG_REQUIRE(ctx && v && keys && df, "null argument");
    const VocabImpl *im = impl_of(v);
    SG_REQUIRE(im != nullptr, "unknown vocab");
    std::vector<int32_t> hdf((size_t)v->n_terms);
    SG_HIP_TRY(hipMemcpyAsync(keys, v->d_keys, sizeof(uint64_t) * (size_t)v->n_terms, hipMemcpyDeviceToHost, ctx->stream));
    SG_HIP_TRY(hipMemcpyAsync(hdf.data(), v->d_df, sizeof(int32_t) * (size_t)v->n_terms, hipMemcpyDeviceToHost, ctx->stream));
    SG_HIP_TRY(hipStreamSynchronize(ctx->stream));
    const int n = v->params.ngram_size, bits = v->bits_per_char;
    for (int64_t i = 0; i < v->n_terms; ++i) {
        df[i] = hdf[(size_t)i];
        // re-pack the compact character codes as 7-bit ASCII, big-endian
        const uint64_t k = keys[i];
        uint64_t out = 0;
        for (int q = 0; q < n; ++q) {
            const uint32_t code = (uint32_t)(k >> (bits * (n - 1 - q))) & ((1u << bits) - 1);
            out = (out << 7) | im->byte_of_rank[code & 127];
        }
        keys[i] = out;
    }
    return SG_OK;
}

extern "C" int sg_vocab_set_idf(sg_ctx *ctx, sg_vocab *v, const void *idf, int32_t dtype) {
    SG_REQUIRE(ctx && v && idf, "null argument");
    SG_REQUIRE(dtype == v->params.dtype, "idf dtype differs from the vectoriser dtype");
    const size_t s = dtype == SG_F64 ? 8 : 4;
    if (!v->d_idf) SG_TRY(ctx->alloc(((size_t)v->n_terms + 1) * s, &v->d_idf));
    SG_HIP_TRY(hipMemcpyAsync(v->d_idf, idf, s * (size_t)v->n_terms, hipMemcpyHostToDevice, ctx->stream));
    SG_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SG_OK;
}

extern "C" int sg_vocab_free(sg_vocab *v) {
    if (!v) return SG_OK;
    VocabImpl *im = v->impl;
    v->impl = nullptr;
    sg_ctx *ctx = v->ctx;
    if (im) {
        for (auto &c : im->caches) free_cache(ctx, c);
        ctx->release(im->d_df_table);
        ctx->release(im->d_err);
        delete im;
    }
    ctx->release(v->d_key_to_col);
    ctx->release(v->d_keys);
    ctx->release(v->d_df);
    ctx->release(v->d_idf);
    delete v;
    return SG_OK;
}

extern "C" int sg_vec_transform(sg_ctx *ctx, const sg_vocab *v, const sg_strings *strings, sg_csr **out) {
    SG_REQUIRE(ctx && v && strings && out, "null argument");
    SG_REQUIRE(v->d_idf != nullptr, "sg_vocab_set_idf has not been called");
    VocabImpl *im = impl_of(v);
    SG_REQUIRE(im != nullptr, "unknown vocab");
    // tokens: reuse the pass made by fit() when these strings were part of it
    TokenCache local;
    const TokenCache *tc = nullptr;
    for (const auto &c : im->caches)
        if (c.src == strings && c.n == strings->n) tc = &c;
    int st = SG_OK;
    if (!tc) {
        SgTimer timer(ctx, SG_K_TOKENIZE);
        const TokParams tp = make_tok_params(v, im);
        (void)hipMemsetAsync(im->d_err, 0, 16, ctx->stream);
        st = tokenize_set(ctx, strings, tp, nullptr, 1, 0, im->d_err, &local);
        if (st != SG_OK) return st;
        tc = &local;
    }
    const int64_t n = strings->n;
    sg_csr *m = new (std::nothrow) sg_csr();
    if (!m) return SG_ERR_OOM;
    m->ctx = ctx;
    m->n_rows = n;
    m->n_cols = v->n_terms;
    m->dtype = v->params.dtype;
    m->owned = true;
    int32_t *kept = nullptr;
    int64_t *indptr = nullptr;
    int32_t host_err = 0;
    int64_t nnz = 0;
    {
        SgTimer timer(ctx, SG_K_WEIGHT);
        st = sg_alloc(ctx, (size_t)n + 1, &kept);
        if (st == SG_OK) st = sg_alloc(ctx, (size_t)n + 1, &indptr);
        if (st == SG_OK && n > 0) {
            hipLaunchKernelGGL(kept_count_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                               (const int64_t *)tc->d_ub_ptr, (const int32_t *)tc->d_cnt,
                               (const uint32_t *)tc->d_keys, (const int32_t *)v->d_key_to_col, n, kept);
            st = sg_exclusive_scan_i32_to_i64(ctx, kept, indptr, n);
        } else if (st == SG_OK) {
            (void)hipMemsetAsync(indptr, 0, sizeof(int64_t), ctx->stream);
        }
        if (st == SG_OK) {
            if (hipMemcpyAsync(&nnz, indptr + n, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                hipMemcpyAsync(&host_err, im->d_err, 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                hipStreamSynchronize(ctx->stream) != hipSuccess)
                st = SG_ERR_HIP;
        }
        if (st == SG_OK && host_err != 0) {
            sg_set_error("a string has more than %d n-grams; the device tokeniser does not handle it", TOK_CAP);
            st = SG_ERR_UNSUPPORTED;
        }
        int32_t *idx = nullptr;
        void *val = nullptr;
        const size_t s = m->dtype == SG_F64 ? 8 : 4;
        if (st == SG_OK) st = sg_alloc(ctx, (size_t)nnz + 4, &idx);
        if (st == SG_OK) st = ctx->alloc(((size_t)nnz + 4) * s, &val);
        m->d_indptr = indptr;
        m->d_indices = idx;
        m->d_data = val;
        m->nnz = nnz;
        if (st == SG_OK && n > 0) {
            const unsigned grid = (unsigned)((n + 255) / 256);
            if (m->dtype == SG_F64)
                hipLaunchKernelGGL(weight_normalize_kernel<double>, dim3(grid), dim3(256), 0, ctx->stream,
                                   (const int64_t *)tc->d_ub_ptr, (const int32_t *)tc->d_cnt,
                                   (const uint32_t *)tc->d_keys, (const int32_t *)tc->d_tf,
                                   (const int32_t *)v->d_key_to_col, (const double *)v->d_idf, n,
                                   (const int64_t *)indptr, idx, (double *)val);
            else
                hipLaunchKernelGGL(weight_normalize_kernel<float>, dim3(grid), dim3(256), 0, ctx->stream,
                                   (const int64_t *)tc->d_ub_ptr, (const int32_t *)tc->d_cnt,
                                   (const uint32_t *)tc->d_keys, (const int32_t *)tc->d_tf,
                                   (const int32_t *)v->d_key_to_col, (const float *)v->d_idf, n,
                                   (const int64_t *)indptr, idx, (float *)val);
            if (hipGetLastError() != hipSuccess) st = SG_ERR_HIP;
        }
    }
    ctx->release(kept);
    if (tc == &local) free_cache(ctx, local);
    if (st != SG_OK) {
        sg_csr_free(m);
        return st;
    }
    *out = m;
    return SG_OK;
}
