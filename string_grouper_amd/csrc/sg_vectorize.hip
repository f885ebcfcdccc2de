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
// Domain (round 2).  Strings of any length: one wave sorts up to 1024 n-grams in LDS, longer strings go to a
// workgroup-per-string kernel with its keys in global scratch.  Keys of up to 30 bits index a dense table (3-grams over
// 7-bit ASCII: 2 M counters); wider ones (long n-grams, large alphabets; up to 63 bits) are 64-bit and the vocabulary
// is the sorted array of the distinct keys (sg_sortvocab.hip), looked up by binary search.  Columns whose n-grams are
// over non-ASCII code points arrive as SYMBOL columns: uint16 ranks in the alphabet of the fit, prepared on the host
// (string_grouper_amd/strprep.py).
//
// Bound: HBM.  Algorithmic bytes = sum(len) + 8n (read strings) + nnz*(4+s) + 8(n+1) (write CSR)
// + the key-space tables (4 * 2^(bits*n) for df and rank).
#include <type_traits>

#include "sg_internal.h"

#define TOK_CAP 1024            // n-grams per string of the one-wave-per-string tokeniser; longer strings: tokenize_long_kernel
#define SG_KEY_OOV32 0xFFFFFFFEu            // key of an n-gram with a character that is not in the fit's alphabet
#define SG_KEY_OOV64 0xFFFFFFFFFFFFFFFEull  // (sorts behind every real key; never in the vocabulary)
#define SG_CHAR_ABSENT 0xFFFFu  // code of such a character
#define TOK_CHARS (TOK_CAP + 16)

template <typename KeyT>
struct KeyTraits;
template <>
struct KeyTraits<uint32_t> {
    static constexpr uint32_t OOV = SG_KEY_OOV32;
    static constexpr uint32_t PAD = 0xFFFFFFFFu;
};
template <>
struct KeyTraits<uint64_t> {
    static constexpr uint64_t OOV = SG_KEY_OOV64;
    static constexpr uint64_t PAD = 0xFFFFFFFFFFFFFFFFull;
};

struct TokenCache {             // tokenised strings in the padded layout
    const sg_strings *src = nullptr;
    int64_t n = 0;
    int64_t cap_total = 0;
    const int64_t *d_ub_ptr = nullptr;   // n + 1: row i's slots start at d_ub_ptr[i] - d_ub_ptr[0] -- round 4: the strings' own offsets
                                         // (borrowed): a row of L characters holds at most L n-grams, and the arrays below are
                                         // sized by the characters anyway; rounds 1-3 counted L - n + 1 per row and scanned
    int32_t *d_cnt = nullptr;      // n: distinct n-grams of row i
    void *d_keys = nullptr;        // cap_total keys (uint32 in dense mode, uint64 in sorted mode)
    int32_t *d_tf = nullptr;       // cap_total
    bool keys_are_columns = false; // d_keys holds column ids instead of n-gram keys (dense mode, after the fit's df pass)
    uint32_t *d_longs = nullptr;   // [0] how many strings are still to be tokenised by the workgroup-per-string kernel, then
                                   // their rows: kept while that question is open (tokenize_set_t, defer_longs)
    // Round 6: the row pointers of the matrix a transform of THESE strings will return (a column of the fit keeps every
    // n-gram: prefix sums of d_cnt), made by the fit's end so that their last entry -- the matrix's non-zeros, which the
    // transform needs on the host to size its arrays -- comes back with the synchronisation the fit makes anyway (the size
    // of the vocabulary); the transform asked on its own before: one of the step's five host round trips.  Handed to the
    // first transform that wants them (its matrix owns them from then on).
    int64_t *d_indptr = nullptr;
    int64_t nnz = -1;
};

struct VocabImpl {
    std::vector<TokenCache> caches;
    uint16_t rank_of_byte[128];    // byte -> compact character code (SG_CHAR_ABSENT: did not occur at fit())
    uint8_t byte_of_rank[128];
    int32_t *d_df_table = nullptr; // dense mode: key_space counters, or marks (df_marks)
    bool df_marks = false;         // the table only marks the keys that occur; df is counted per column at fit_end
    int64_t df_stride = 0;         // entries of one copy of the table
    bool local_alphabet = false;   // byte columns coded by rank among the bytes seen at fit() (7 * ngram_size > 24)
    bool symbols = false;          // the fit's columns are symbol columns (sg_strings_from_host_symbols)
    int32_t alphabet = 0;          // their alphabet size
};

static VocabImpl *impl_of(const sg_vocab *v) { return v ? v->impl : nullptr; }

struct TokParams {
    int32_t ngram;
    int32_t lower;                 // byte columns: map A-Z to a-z
    int32_t bits;
    uint32_t del_mask[4];          // bit c set: ASCII byte c is deleted
    uint16_t rank_of_byte[128];
};

// -------------------------------------------------------------------------------------------------
// alphabet presence: which filtered byte values occur at all (only needed when 7*n bits is too wide)
__global__ void __launch_bounds__(256) alphabet_kernel(const uint8_t *__restrict__ bytes, int64_t total, TokParams p,
                                                       uint32_t *present /*[4]*/) {
    __shared__ uint32_t local[4];
    if (threadIdx.x < 4) local[threadIdx.x] = 0;
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t c = bytes[i];
        if (c >= 0x80) continue;
        if (p.lower && c >= 'A' && c <= 'Z') c += 32;
        if ((p.del_mask[c >> 5] >> (c & 31)) & 1u) continue;
        atomicOr(&local[c >> 5], 1u << (c & 31));
    }
    __syncthreads();
    if (threadIdx.x < 4 && local[threadIdx.x]) atomicOr(&present[threadIdx.x], local[threadIdx.x]);
}

// The character code of a byte (byte columns: filter, lower, delete, rank) or of a symbol (symbol columns: as it is).
// Returns false when the character is dropped.
template <bool SYMBOLS>
__device__ __forceinline__ bool char_code(const void *chars, int64_t at, const TokParams &p, uint32_t *code) {
    if (SYMBOLS) {
        *code = reinterpret_cast<const uint16_t *>(chars)[at];
        return true;
    }
    uint32_t c = reinterpret_cast<const uint8_t *>(chars)[at];
    if (p.lower && c >= 'A' && c <= 'Z') c += 32;
    if (c >= 0x80 || ((p.del_mask[(c & 127) >> 5] >> (c & 31)) & 1u)) return false;
    *code = p.rank_of_byte[c];
    return true;
}

// One wave per string: filter, n-gram, sort, run-length encode.
// The tokeniser runs single-wave workgroups: LDS operations of one wave execute in issue order, so its
// phases need no s_barrier (and no wait for the outstanding df atomics / token stores of the previous
// string, which __syncthreads() implies); a compiler-level barrier keeps the LDS accesses in program order.
__device__ __forceinline__ void wave_sync() { __builtin_amdgcn_wave_barrier(); }

// Document frequencies.  Two forms:
//  * counters (df_replicas >= 1): one device-scope atomic per (row, distinct n-gram) into a dense table over the key
//    space -- the form whose table a multi-GPU caller sums across ranks (sg_vocab_df_table).  12.5 M atomics at 663 k
//    names cost 0.85 ms, twice the rest of the tokeniser (profiles/r02_sessionK_k1_probe.log).
//  * marks (df_replicas == 0; the single-GPU sg_vec_fit): the tokeniser only marks the keys that occur -- a load that
//    hits in L1/L2 and, for the first rows that see a key, a plain store of 1 (racing stores all write the same value) --
//    and the counts are made afterwards per COLUMN in LDS (df_count_lds_kernel), without any global atomic.
__device__ __forceinline__ void count_document(int32_t *df_table, int32_t df_replicas, int64_t df_stride, int64_t row,
                                               int64_t key) {
    if (df_replicas == 0) {
        if (df_table[key] == 0) df_table[key] = 1;
    } else {
        atomicAdd(&df_table[(int64_t)(row % df_replicas) * df_stride + key], 1);
    }
}

// The common case -- a string of at most 64 characters -- one wave per string, four strings per workgroup, without the
// LDS sorting network: every lane holds one n-gram key and finds its rank by comparing it with all the others, which
// are broadcast from lane to lane through scalar registers (v_readlane: no memory, no dependent LDS round trips; an LDS
// bitonic sort of 64 keys is 21 dependent read/compare/write stages).  Ties are broken by position so that the ranks
// are a permutation; the keys are scattered to their ranks in LDS once, read back in order, and run-length encoded with
// one ballot.  Longer strings are queued for tokenize_kernel.
template <typename KeyT>
__device__ __forceinline__ KeyT read_lane_key(KeyT v, int i);
template <>
__device__ __forceinline__ uint32_t read_lane_key<uint32_t>(uint32_t v, int i) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, i);
}
template <>
__device__ __forceinline__ uint64_t read_lane_key<uint64_t>(uint64_t v, int i) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, i);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), i);
    return ((uint64_t)hi << 32) | lo;
}

#define TOK_SHORT_WAVES 4
template <typename KeyT, bool SYMBOLS>
__global__ void __launch_bounds__(64 * TOK_SHORT_WAVES) tokenize_short_kernel(
    const void *__restrict__ chars_in, const int64_t *__restrict__ offsets, int64_t n_rows, TokParams p,
    const int64_t *__restrict__ ub_ptr, int32_t *__restrict__ out_cnt, KeyT *__restrict__ out_keys,
    int32_t *__restrict__ out_tf, int32_t *df_table, int32_t df_replicas, int64_t df_stride, uint32_t *mid_count,
    uint32_t *mid_rows) {
    __shared__ KeyT skeys[TOK_SHORT_WAVES][64];
    __shared__ uint16_t schars[TOK_SHORT_WAVES][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    KeyT *keys = skeys[wv];
    uint16_t *chars = schars[wv];
    const uint64_t lt_mask = ((uint64_t)1 << lane) - 1;
    // keys that leave six spare bits are ranked with the lane number appended: one comparison per pair
    const bool tagged = sizeof(KeyT) == 4 && p.bits * p.ngram <= 25;
    // A wave works off ~80 strings one after the other, and a string was a chain of dependent loads -- offsets, characters,
    // the row's slot, the document-frequency mark -- each a round trip of a microsecond or two under load: the kernel ran
    // at the latency of its loads (0.39 ms for 16 MB in, 130 MB out at 663 k).  Round 4: a software pipeline.
    //   * A wave takes BLOCKS of sixteen consecutive strings: their offsets and slots are two coalesced loads, handed
    //     to the strings lane to lane (v_readlane) -- no load per string for them.
    //   * The characters of string i + 1 are loaded while string i is tokenised; the two live in registers of their
    //     own (the trip is written twice, A and B): rotating ONE register would be a copy of a register whose load is
    //     in flight, i.e. a wait.
    //   * What a trip computes -- keys, counts, marks -- is stored at the head of the NEXT trip: the wait counter
    //     covers stores as well as loads, in order, and the compiler waits for "everything" at a trip's first use of a
    //     loaded value; that way everything is a whole trip old.
    //   * Inside a trip nothing else loads from memory: the tables of the kernel's argument block, indexed per lane,
    //     would be served with loads from the block's memory -- the deletion mask is selected from its four words, the
    //     character ranks are read from a copy in LDS, and a key's mark is a plain store (count_document: racing
    //     stores write the same 1).
    __shared__ uint16_t s_rank[128];
    if (threadIdx.x < 128) s_rank[threadIdx.x] = p.rank_of_byte[threadIdx.x];
    __syncthreads();
    // (two 64-bit words and one select: a four-way select on the words the compiler turns into a table in scratch memory)
    const uint64_t dm_lo = (uint64_t)p.del_mask[0] | ((uint64_t)p.del_mask[1] << 32), dm_hi = (uint64_t)p.del_mask[2] | ((uint64_t)p.del_mask[3] << 32);
    struct Pending {
        bool head;        // this lane holds a distinct n-gram
        bool mark;        // ... that is in the vocabulary's key space
        int64_t at;       // its slot
        KeyT key;
        int32_t tf;
        int64_t row;      // the row whose count goes out (-1: none)
        int32_t cnt;
    };
    auto store_pending = [&](const Pending &q) {
        if (q.head) {
            out_keys[q.at] = q.key;
            out_tf[q.at] = q.tf;
            if (q.mark) {
                if (df_replicas == 0) df_table[(int64_t)q.key] = 1;   // mark (see count_document): racing stores write the same 1
                else count_document(df_table, df_replicas, df_stride, q.row, (int64_t)q.key);
            }
        }
        if (lane == 0 && q.row >= 0) out_cnt[q.row] = q.cnt;
    };
    auto read_lane_i64 = [&](int64_t v, int i) -> int64_t {
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(uint64_t)v, i);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)v >> 32), i);
        return (int64_t)(((uint64_t)hi << 32) | lo);
    };
    auto load_raw = [&](int64_t b0, int64_t b1) -> uint32_t {   // the lane's character, untouched (0 past the end / for long rows)
        const int64_t len = b1 - b0;
        if (len > 64 || lane >= len) return 0u;
        return SYMBOLS ? (uint32_t)reinterpret_cast<const uint16_t *>(chars_in)[b0 + lane]
                       : (uint32_t)reinterpret_cast<const uint8_t *>(chars_in)[b0 + lane];
    };
    // the character code of the lane's character (filter, lower, delete, rank): false = dropped
    auto code_of = [&](uint32_t raw, int64_t len, uint32_t &code) -> bool {
        code = 0;
        if (lane >= len) return false;
        if (SYMBOLS) {
            code = raw;
            return true;
        }
        uint32_t c = raw;
        if (p.lower && c >= 'A' && c <= 'Z') c += 32;
        const uint64_t dm = c < 64 ? dm_lo : dm_hi;
        if (c >= 0x80 || ((dm >> (c & 63)) & 1ull)) return false;
        code = s_rank[c];
        return true;
    };
    // one string, its characters coded: n-grams, ranks, run lengths -> what the next trip stores
    auto tokenize_row = [&](int64_t row, int64_t len, int64_t ub, bool keep, uint32_t code) -> Pending {
        Pending q{false, false, 0, 0, 0, row, 0};
        if (len > 64) {
            if (lane == 0) mid_rows[atomicAdd(mid_count, 1u)] = (uint32_t)row;   // (its count, 0, goes out with the next trip)
            return q;
        }
        const uint64_t km = __ballot(keep);
        const int g = __popcll(km) - p.ngram + 1;   // number of n-grams
        if (g <= 0) return q;
        if (keep) chars[__popcll(km & lt_mask)] = (uint16_t)code;
        wave_sync();
        KeyT key = KeyTraits<KeyT>::PAD;
        if (lane < g) {
            key = 0;
            bool absent = false;
            for (int e = 0; e < p.ngram; ++e) {
                const uint32_t ch = chars[lane + e];
                absent |= ch == SG_CHAR_ABSENT;
                key = (key << p.bits) | (KeyT)ch;
            }
            if (absent) key = KeyTraits<KeyT>::OOV;
        }
        uint32_t rank = 0;
        if (tagged) {
            const uint32_t body = key == KeyTraits<KeyT>::OOV ? 0x3FFFFFFu : (uint32_t)key;   // real keys are < 2^25
            const uint32_t kt = lane < g ? (body << 6) | (uint32_t)lane : 0xFFFFFFFFu;
            // four comparisons per trip of the loop (a lane past the last n-gram holds the largest value: it counts for
            // nobody): the loop's own scalar instructions were as many as its vector ones, and the kernel is bound by the
            // instructions it issues since its loads are pipelined
            const int g4 = (g + 3) & ~3;
#pragma unroll 4
            for (int i = 0; i < g4; ++i) rank += (uint32_t)__builtin_amdgcn_readlane((int)kt, i) < kt;
        } else {
            for (int i = 0; i < g; ++i) {
                const KeyT o = read_lane_key<KeyT>(key, i);
                rank += (o < key) || (o == key && i < lane);
            }
        }
        if (lane < g) keys[rank] = key;
        wave_sync();
        bool head = false;
        KeyT mine = 0;
        if (lane < g) {
            mine = keys[lane];
            head = lane == 0 || mine != keys[lane - 1];
        }
        const uint64_t hm = __ballot(head);
        if (head) {
            const uint64_t above = (hm >> lane) >> 1;   // the heads behind this one
            const int next = above ? lane + 1 + __builtin_ctzll(above) : g;
            q.head = true;
            q.at = ub + __popcll(hm & lt_mask);
            q.key = mine;
            q.tf = next - lane;
            q.mark = df_table != nullptr && mine != KeyTraits<KeyT>::OOV;
        }
        q.cnt = __popcll(hm);
        wave_sync();   // the next string overwrites chars / keys
        return q;
    };
    constexpr int BLK = 16;
    const int64_t n_blocks = (n_rows + BLK - 1) / BLK;
    const int64_t stride = (int64_t)gridDim.x * TOK_SHORT_WAVES;
    Pending pend{false, false, 0, 0, 0, -1, 0};
    for (int64_t blk = (int64_t)blockIdx.x * TOK_SHORT_WAVES + wv; blk < n_blocks; blk += stride) {
        const int64_t r0 = blk * BLK;
        const int nrow = (int)min((int64_t)BLK, n_rows - r0);
        // lane l <= nrow: the l-th offset of the block; lane l < nrow: the slot of its l-th string
        const int64_t m_off = lane <= nrow ? offsets[r0 + lane] : 0;
        const int64_t m_ub = lane < nrow ? ub_ptr[r0 + lane] - ub_ptr[0] : 0;
        int64_t b0A = read_lane_i64(m_off, 0), b1A = read_lane_i64(m_off, 1);
        uint32_t rawA = load_raw(b0A, b1A), rawB = 0;
        for (int i = 0; i < BLK; i += 2) {   // (BLK even; strings past the block's end have no characters and no row)
            {   // trip A: string i
                uint32_t code;
                const int64_t len = i < nrow ? b1A - b0A : 0;
                const bool keep = code_of(rawA, len, code);
                const int64_t b0B = i + 1 < nrow ? read_lane_i64(m_off, i + 1) : 0, b1B = i + 1 < nrow ? read_lane_i64(m_off, i + 2) : 0;
                rawB = load_raw(b0B, b1B);
                store_pending(pend);
                pend = i < nrow ? tokenize_row(r0 + i, len, read_lane_i64(m_ub, i), keep, code) : Pending{false, false, 0, 0, 0, -1, 0};
                b0A = b0B;   // (scalars: no load is in flight into them)
                b1A = b1B;
            }
            {   // trip B: string i + 1 (b0A, b1A are its offsets now)
                uint32_t code;
                const int64_t len = i + 1 < nrow ? b1A - b0A : 0;
                const bool keep = code_of(rawB, len, code);
                const int64_t b0N = i + 2 < nrow ? read_lane_i64(m_off, i + 2) : 0, b1N = i + 2 < nrow ? read_lane_i64(m_off, i + 3) : 0;
                rawA = load_raw(b0N, b1N);
                store_pending(pend);
                pend = i + 1 < nrow ? tokenize_row(r0 + i + 1, len, read_lane_i64(m_ub, i + 1), keep, code) : Pending{false, false, 0, 0, 0, -1, 0};
                b0A = b0N;
                b1A = b1N;
            }
        }
    }
    store_pending(pend);
}

template <typename KeyT, bool SYMBOLS>
__global__ void __launch_bounds__(64) tokenize_kernel(const void *__restrict__ chars_in,
                                                      const int64_t *__restrict__ offsets, int64_t n_rows,
                                                      TokParams p, const int64_t *__restrict__ ub_ptr,
                                                      int32_t *__restrict__ out_cnt, KeyT *__restrict__ out_keys,
                                                      int32_t *__restrict__ out_tf, int32_t *df_table,
                                                      int32_t df_replicas, int64_t df_stride,
                                                      uint32_t *long_count, uint32_t *long_rows,
                                                      const uint32_t *__restrict__ row_list /* the rows to do, or null: all */,
                                                      const uint32_t *__restrict__ row_list_len) {
    __shared__ KeyT keys[TOK_CAP];
    __shared__ uint16_t starts[TOK_CAP + 2];
    __shared__ uint16_t chars[TOK_CHARS];
    const int lane = threadIdx.x;
    const uint64_t lt_mask = ((uint64_t)1 << lane) - 1;

    const int64_t n_todo = row_list ? (int64_t)*row_list_len : n_rows;
    for (int64_t at = blockIdx.x; at < n_todo; at += gridDim.x) {
        const int64_t row = row_list ? (int64_t)row_list[at] : at;
        const int64_t b0 = offsets[row];
        const int64_t len = offsets[row + 1] - b0;
        // ---- filter + compact into LDS
        int m = 0;   // surviving characters
        bool overflow = false;
        for (int64_t base = 0; base < len; base += 64) {
            uint32_t code = 0;
            const bool keep = base + lane < len && char_code<SYMBOLS>(chars_in, b0 + base + lane, p, &code);
            const uint64_t km = __ballot(keep);
            const int pos = m + __popcll(km & lt_mask);
            if (keep && pos < TOK_CHARS) chars[pos] = (uint16_t)code;
            m += __popcll(km);
            if (m > TOK_CHARS) {
                overflow = true;
                break;
            }
        }
        wave_sync();
        int g = m - p.ngram + 1;   // number of n-grams
        if (g < 0) g = 0;
        if (overflow || g > TOK_CAP) {   // a long string: the workgroup-per-string kernel takes it
            if (lane == 0) {
                long_rows[atomicAdd(long_count, 1u)] = (uint32_t)row;
                out_cnt[row] = 0;
            }
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
            KeyT key = KeyTraits<KeyT>::PAD;
            if (idx < g) {
                key = 0;
                bool absent = false;   // a character that is not in the alphabet of fit(): the n-gram is out of vocabulary
                for (int q = 0; q < p.ngram; ++q) {
                    const uint32_t ch = chars[idx + q];
                    absent |= ch == SG_CHAR_ABSENT;
                    key = (key << p.bits) | (KeyT)ch;
                }
                if (absent) key = KeyTraits<KeyT>::OOV;
            }
            keys[idx] = key;
        }
        wave_sync();
        // ---- bitonic sort, ascending
        for (int k = 2; k <= P; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int idx = lane; idx < (P >> 1); idx += 64) {
                    const int i = ((idx & ~(j - 1)) << 1) | (idx & (j - 1));
                    const KeyT a = keys[i], b = keys[i + j];
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
        const int64_t obase = ub_ptr[row] - ub_ptr[0];
        for (int u = lane; u < uniq; u += 64) {
            const int s0 = starts[u];
            const KeyT key = keys[s0];
            out_keys[obase + u] = key;
            out_tf[obase + u] = (int32_t)starts[u + 1] - s0;
            // the same few n-grams ('inc', ' co') occur in a fifth of all strings: spread their atomics over
            // several copies of the table (summed by df_reduce_kernel) instead of serialising on one address
            if (df_table && key != KeyTraits<KeyT>::OOV) count_document(df_table, df_replicas, df_stride, row, (int64_t)key);
        }
        if (lane == 0) out_cnt[row] = uniq;
        wave_sync();
    }
}

// A string with more n-grams than one wave sorts in LDS (an address line, a description, a whole document): one
// 256-thread workgroup per string, characters and keys in global scratch (L2-resident), bitonic sort with workgroup
// barriers.  Rare, so it favours simplicity over speed.  scratch layout per long string e: chars at
// lchars + coff[e] (its byte / symbol length), keys at lkeys + koff[e] (the next power of two of that length).
template <typename KeyT, bool SYMBOLS>
__global__ void __launch_bounds__(256) tokenize_long_kernel(const void *__restrict__ chars_in,
                                                            const int64_t *__restrict__ offsets, TokParams p,
                                                            const uint32_t *__restrict__ long_rows,
                                                            const int64_t *__restrict__ coff, const int64_t *__restrict__ koff,
                                                            uint16_t *lchars, KeyT *lkeys,
                                                            const int64_t *__restrict__ ub_ptr, int32_t *__restrict__ out_cnt,
                                                            KeyT *__restrict__ out_keys, int32_t *__restrict__ out_tf,
                                                            int32_t *df_table, int32_t df_replicas, int64_t df_stride) {
    __shared__ int32_t wave_tot[4];
    __shared__ int32_t carry;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t row = long_rows[blockIdx.x];
    const int64_t b0 = offsets[row];
    const int64_t len = offsets[row + 1] - b0;
    uint16_t *ch = lchars + coff[blockIdx.x];
    KeyT *keys = lkeys + koff[blockIdx.x];
    const int64_t P = koff[blockIdx.x + 1] - koff[blockIdx.x];
    if (tid == 0) carry = 0;
    __syncthreads();
    // ---- filter + compact (block-wide, 256 characters per step)
    for (int64_t base = 0; base < len; base += 256) {
        uint32_t code = 0;
        const bool keep = base + tid < len && char_code<SYMBOLS>(chars_in, b0 + base + tid, p, &code);
        const uint64_t km = __ballot(keep);
        if (lane == 0) wave_tot[wv] = __popcll(km);
        __syncthreads();
        int before = carry;
        for (int w = 0; w < wv; ++w) before += wave_tot[w];
        if (keep) ch[before + __popcll(km & (((uint64_t)1 << lane) - 1))] = (uint16_t)code;
        __syncthreads();
        if (tid == 0) carry += wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
        __syncthreads();
    }
    const int64_t m = carry;
    int64_t g = m - p.ngram + 1;
    if (g < 0) g = 0;
    // ---- keys, padded to the power of two
    for (int64_t idx = tid; idx < P; idx += 256) {
        KeyT key = KeyTraits<KeyT>::PAD;
        if (idx < g) {
            key = 0;
            bool absent = false;
            for (int q = 0; q < p.ngram; ++q) {
                const uint32_t c = ch[idx + q];
                absent |= c == SG_CHAR_ABSENT;
                key = (key << p.bits) | (KeyT)c;
            }
            if (absent) key = KeyTraits<KeyT>::OOV;
        }
        keys[idx] = key;
    }
    __syncthreads();
    for (int64_t k = 2; k <= P; k <<= 1) {
        for (int64_t j = k >> 1; j > 0; j >>= 1) {
            for (int64_t idx = tid; idx < (P >> 1); idx += 256) {
                const int64_t i = ((idx & ~(j - 1)) << 1) | (idx & (j - 1));
                const KeyT a = keys[i], b = keys[i + j];
                const bool up = (i & k) == 0;
                if ((a > b) == up) {
                    keys[i] = b;
                    keys[i + j] = a;
                }
            }
            __syncthreads();
        }
    }
    // ---- run-length encode into the string's slots (block-wide, 256 keys per step)
    const int64_t obase = ub_ptr[row] - ub_ptr[0];
    __syncthreads();
    if (tid == 0) carry = 0;   // distinct keys so far
    __syncthreads();
    for (int64_t base = 0; base < g; base += 256) {
        const int64_t idx = base + tid;
        const bool head = idx < g && (idx == 0 || keys[idx] != keys[idx - 1]);
        const uint64_t hm = __ballot(head);
        if (lane == 0) wave_tot[wv] = __popcll(hm);
        __syncthreads();
        int before = carry;
        for (int w = 0; w < wv; ++w) before += wave_tot[w];
        if (head) {
            const int64_t u = before + __popcll(hm & (((uint64_t)1 << lane) - 1));
            const KeyT key = keys[idx];
            int64_t e = idx + 1;      // the run's end: a scan forward (a run = the occurrences of one n-gram in one string)
            while (e < g && keys[e] == key) ++e;
            out_keys[obase + u] = key;
            out_tf[obase + u] = (int32_t)(e - idx);
            if (df_table && key != KeyTraits<KeyT>::OOV) count_document(df_table, df_replicas, df_stride, row, (int64_t)key);
        }
        __syncthreads();
        if (tid == 0) carry += wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
        __syncthreads();
    }
    if (tid == 0) out_cnt[row] = carry;
}

__global__ void __launch_bounds__(256) long_sizes_kernel(const int64_t *__restrict__ offsets, const uint32_t *__restrict__ long_rows,
                                                         uint32_t n_long, int64_t *sizes) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n_long) sizes[e] = offsets[long_rows[e] + 1] - offsets[long_rows[e]];
}

__global__ void __launch_bounds__(256) df_reduce_kernel(int32_t *df_table, int64_t key_space, int32_t replicas, int64_t stride) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= key_space) return;
    int32_t s = df_table[i];
    for (int r = 1; r < replicas; ++r) s += df_table[(int64_t)r * stride + i];
    df_table[i] = s;
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


// df of every column, marks form: a workgroup walks a block of rows (sixteen lanes per row: the token loads are
// contiguous), counts the columns of its tokens in an LDS histogram over [col0, col0 + n_cols) and stores the histogram
// as one row of `partial`; df_sum_kernel adds the rows up.
__global__ void __launch_bounds__(1024) df_count_lds_kernel(const int64_t *__restrict__ ub_ptr, const int32_t *__restrict__ cnt,
                                                            uint32_t *keys /* read; with cols_out also written */,
                                                            const int32_t *__restrict__ key_to_col, int64_t n_rows,
                                                            int32_t col0, int32_t n_cols, int64_t n_terms,
                                                            uint32_t *__restrict__ partial,
                                                            int32_t cols_out /* != 0: every key is REPLACED by its column (the pass
                                                                                over all columns of a fit's own tokens): the weighting
                                                                                kernel then has no table to gather from -- 12.5 M reads
                                                                                scattered over an 8 MB table at 663 k were its cost */) {
    extern __shared__ uint32_t df_hist[];
    for (int k = threadIdx.x; k < n_cols; k += blockDim.x) df_hist[k] = 0;
    __syncthreads();
    const int64_t per_wg = (n_rows + gridDim.x - 1) / gridDim.x;
    const int64_t r0 = (int64_t)blockIdx.x * per_wg;
    int64_t r1 = r0 + per_wg;
    if (r1 > n_rows) r1 = n_rows;
    const int sub = threadIdx.x & 15;
    const int64_t groups = blockDim.x >> 4;
    // four rows per sixteen lanes and trip, every load of a stage issued before the first is used (a workgroup's trips
    // are serial: the kernel ran at the latency of its three dependent loads per trip)
    for (int64_t rb = r0 + (threadIdx.x >> 4) * 4; rb < r1; rb += groups * 4) {
        int64_t b[4];
        int c[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool valid = rb + r < r1;
            b[r] = valid ? ub_ptr[rb + r] - ub_ptr[0] : 0;
            c[r] = valid ? cnt[rb + r] : 0;
        }
        uint32_t key[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) key[r] = sub < c[r] ? keys[b[r] + sub] : SG_KEY_OOV32;
        int32_t col[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) col[r] = key[r] != SG_KEY_OOV32 ? key_to_col[key[r]] : -1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (sub < c[r]) {
                if (cols_out) keys[b[r] + sub] = (uint32_t)col[r];
                const int32_t cc = col[r] - col0;
                if (col[r] >= 0 && cc >= 0 && cc < n_cols) atomicAdd(&df_hist[cc], 1u);
            }
            for (int q = sub + 16; q < c[r]; q += 16) {          // rows of more than sixteen distinct n-grams
                const uint32_t k2 = keys[b[r] + q];
                const int32_t c2 = k2 != SG_KEY_OOV32 ? key_to_col[k2] : -1;
                if (cols_out) keys[b[r] + q] = (uint32_t)c2;
                const int32_t cc = c2 - col0;
                if (c2 >= 0 && cc >= 0 && cc < n_cols) atomicAdd(&df_hist[cc], 1u);
            }
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < n_cols; k += blockDim.x) partial[(int64_t)blockIdx.x * n_terms + col0 + k] = df_hist[k];
}

__global__ void __launch_bounds__(256) df_sum_kernel(const uint32_t *__restrict__ partial, int32_t n_partial, int64_t n_terms,
                                                     int32_t *__restrict__ df) {
    // 64 columns per workgroup, four threads per column (each a quarter of the workgroups' histograms, joined through LDS);
    // loads are contiguous across the 64 columns.  The sum is WRITTEN: no clear of df first (round 3: atomics into a
    // cleared array).
    __shared__ uint32_t part[4][64];
    const int kk = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int64_t k = (int64_t)blockIdx.x * 64 + kk;
    const int per = (n_partial + 3) / 4;
    const int r0 = q * per, r1 = r0 + per < n_partial ? r0 + per : n_partial;
    uint32_t s = 0;
    if (k < n_terms)
        for (int r = r0; r < r1; ++r) s += partial[(int64_t)r * n_terms + k];
    part[q][kk] = s;
    __syncthreads();
    if (q == 0 && k < n_terms) df[k] = (int32_t)(part[0][kk] + part[1][kk] + part[2][kk] + part[3][kk]);
}

// column of a key: dense mode = table lookup, sorted mode = binary search in the ascending vocabulary
struct DenseLookup {
    const int32_t *key_to_col;
    __device__ __forceinline__ int32_t operator()(uint32_t key) const { return key == SG_KEY_OOV32 ? -1 : key_to_col[key]; }
};
// the keys of the cache ARE columns already (df_count_lds_kernel, cols_out; -1 = out of vocabulary)
struct ColumnsLookup {
    __device__ __forceinline__ int32_t operator()(uint32_t key) const { return (int32_t)key; }
};
struct SortedLookup {
    const uint64_t *vocab;
    int64_t n_terms;
    __device__ __forceinline__ int32_t operator()(uint64_t key) const {
        int64_t lo = 0, hi = n_terms;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (vocab[mid] < key) lo = mid + 1;
            else hi = mid;
        }
        return (lo < n_terms && vocab[lo] == key) ? (int32_t)lo : -1;
    }
};

template <typename KeyT, typename Lookup>
__global__ void __launch_bounds__(256) kept_count_kernel(const int64_t *__restrict__ ub_ptr,
                                                         const int32_t *__restrict__ cnt,
                                                         const KeyT *__restrict__ keys, Lookup lookup, int64_t n,
                                                         int32_t *kept) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t b = ub_ptr[i] - ub_ptr[0];
    const int c = cnt[i];
    int k = 0;
    for (int q = 0; q < c; ++q) k += lookup(keys[b + q]) >= 0;
    kept[i] = k;
}

// all (row, distinct n-gram) keys of a token cache, back to back (input of the sorted vocabulary)
__global__ void __launch_bounds__(256) gather_keys_kernel(const int64_t *__restrict__ ub_ptr, const int32_t *__restrict__ cnt,
                                                          const uint64_t *__restrict__ keys, const int64_t *__restrict__ dst_ptr,
                                                          int64_t n, uint64_t *__restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t b = ub_ptr[i] - ub_ptr[0], d = dst_ptr[i];
    const int c = cnt[i];
    for (int q = 0; q < c; ++q) dst[d + q] = keys[b + q];
}

template <typename T>
__device__ __forceinline__ T tmul(T a, T b);
template <>
__device__ __forceinline__ float tmul<float>(float a, float b) { return __fmul_rn(a, b); }
template <>
__device__ __forceinline__ double tmul<double>(double a, double b) { return __dmul_rn(a, b); }

template <typename T, typename KeyT, typename Lookup>
__global__ void __launch_bounds__(256) weight_normalize_kernel(const int64_t *__restrict__ ub_ptr,
                                                               const int32_t *__restrict__ cnt,
                                                               const KeyT *__restrict__ keys,
                                                               const int32_t *__restrict__ tf, Lookup lookup,
                                                               const T *__restrict__ idf, int64_t n,
                                                               const int64_t *__restrict__ indptr,
                                                               int32_t *__restrict__ out_idx, T *__restrict__ out_val,
                                                               uint32_t *props /* [1] max ||row||^2 (float bits) [2] longest row */) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float n2 = 0.f;
    uint32_t len = 0;
    if (i < n) {
        const int64_t b = ub_ptr[i] - ub_ptr[0];
        const int c = cnt[i];
        int64_t o = indptr[i];
        const int64_t o0 = o;
        double acc = 0.0;
        for (int q = 0; q < c; ++q) {   // ascending key == ascending column
            const int32_t col = lookup(keys[b + q]);
            if (col < 0) continue;       // out-of-vocabulary n-gram of a string that was not part of fit()
            const T w = tmul<T>((T)tf[b + q], idf[col]);
            out_idx[o] = col;
            out_val[o] = w;
            acc = __dadd_rn(acc, (double)tmul<T>(w, w));
            ++o;
        }
        len = (uint32_t)(o - o0);
        if (acc != 0.0) {
            const double nrm = __dsqrt_rn(acc);
            double s2 = 0.0;   // norm^2 of the row as stored (what sg_csr_props would compute by scanning the matrix)
            for (int64_t q = o0; q < o; ++q) {
                const T v = (T)__ddiv_rn((double)out_val[q], nrm);
                out_val[q] = v;
                s2 += (double)v * (double)v;
            }
            n2 = __double2float_ru(s2);
        }
    }
    // the multiply's pruning rules need the largest row norm and the longest row: reduce them here, where the rows are made
    uint32_t nb = __float_as_uint(n2);   // non-negative floats order like unsigned integers
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        nb = max(nb, (uint32_t)__shfl_xor((int)nb, d, 64));
        len = max(len, (uint32_t)__shfl_xor((int)len, d, 64));
    }
    // one counter takes ~12 ns per atomic whoever sends it: look first (a stale value is lower, never higher)
    if ((threadIdx.x & 63) == 0 && props) {
        const volatile uint32_t *seen = props;
        if (nb > seen[1]) atomicMax(props + 1, nb);
        if (len > seen[2]) atomicMax(props + 2, len);
    }
}

// K2, the form that runs: sixteen lanes per row (four rows per wave), so that the token loads and the matrix stores of a
// row are contiguous across lanes -- a thread per row (weight_normalize_kernel above, kept as the plain statement of the
// arithmetic) walks 64 rows with a stride of one row per lane and misses in L1 on every step.
// The sum of squares must be added in column order in double (sklearn's loop; the roundings of an f64 row depend on the
// order): it runs down the sixteen lanes as a chain of row_shr:1 DPP moves -- after step k lane k holds the sum of the
// tokens up to its own, lane 0 re-adds the carry of the previous sixteen tokens every step -- one add per token and no
// memory.  The second pass recomputes the weights (the loads hit in L1) and writes col / w / ||row|| at the compacted
// position.  Skipped (out-of-vocabulary) tokens add 0.0, which leaves the sum as it is.
__device__ __forceinline__ double dpp_from_lower_lane(double v, double lane0_gets) {
    const uint64_t b = (uint64_t)__double_as_longlong(v), o = (uint64_t)__double_as_longlong(lane0_gets);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)o, (int)(uint32_t)b, 0x111 /* row_shr:1 */, 0xF, 0xF, false);
    const uint32_t hi =
        (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)(o >> 32), (int)(uint32_t)(b >> 32), 0x111, 0xF, 0xF, false);
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}
__device__ __forceinline__ double shfl_double(double v, int src_lane) {
    const uint64_t b = (uint64_t)__double_as_longlong(v);
    const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)b, src_lane, 64), hi = (uint32_t)__shfl((int)(uint32_t)(b >> 32), src_lane, 64);
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}

template <typename T, typename KeyT, typename Lookup>
__global__ void __launch_bounds__(256) weight_normalize_rows16_kernel(const int64_t *__restrict__ ub_ptr,
                                                                      const int32_t *__restrict__ cnt,
                                                                      const KeyT *__restrict__ keys,
                                                                      const int32_t *__restrict__ tf, Lookup lookup,
                                                                      const T *__restrict__ idf, int64_t n,
                                                                      const int64_t *__restrict__ indptr,
                                                                      int32_t *__restrict__ out_idx, T *__restrict__ out_val,
                                                                      uint32_t *props) {
    const int lane = threadIdx.x & 63, sub = lane & 15, grp_shift = lane & 48;
    const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const bool valid = row < n;
    const int64_t b = valid ? ub_ptr[row] - ub_ptr[0] : 0;
    const int c = valid ? cnt[row] : 0;
    const int64_t o0 = valid ? indptr[row] : 0;
    int cmax = c;   // the wave walks its four rows together
    cmax = max(cmax, __shfl_xor(cmax, 16, 64));
    cmax = max(cmax, __shfl_xor(cmax, 32, 64));
    double carry = 0.0;
    // (round 6: the columns and weights of the first two trips -- 32 entries, nearly every row of a name list -- stay in
    //  registers for the second pass, which looked every key up and fetched tf and idf again)
    int32_t col_keep[2] = {-1, -1};
    T w_keep[2] = {(T)0, (T)0};
    for (int base = 0; base < cmax; base += 16) {
        const int q = base + sub;
        double w2 = 0.0;
        if (q < c) {
            const int32_t col = lookup(keys[b + q]);
            if (col >= 0) {
                const T w = tmul<T>((T)tf[b + q], idf[col]);
                w2 = (double)tmul<T>(w, w);
                if (base == 0) {
                    col_keep[0] = col;
                    w_keep[0] = w;
                } else if (base == 16) {
                    col_keep[1] = col;
                    w_keep[1] = w;
                }
            }
        }
        double acc = w2;
#pragma unroll
        for (int k = 0; k < 16; ++k) acc = __dadd_rn(dpp_from_lower_lane(acc, carry), w2);
        carry = shfl_double(acc, lane | 15);
    }
    float n2 = 0.f;
    uint32_t len = 0;
    if (carry != 0.0) {
        const double nrm = __dsqrt_rn(carry);
        double s2 = 0.0;
        int kept_before = 0;
        for (int base = 0; base < cmax; base += 16) {
            const int q = base + sub;
            int32_t col = -1;
            T w = 0;
            if (base < 32) {          // (kept by the first pass; -1 / 0 for a lane beyond the row or a key without a column)
                col = base == 0 ? col_keep[0] : col_keep[1];
                w = base == 0 ? w_keep[0] : w_keep[1];
            } else if (q < c) {
                col = lookup(keys[b + q]);
                if (col >= 0) w = tmul<T>((T)tf[b + q], idf[col]);
            }
            const uint32_t km = (uint32_t)(__ballot(col >= 0) >> grp_shift) & 0xFFFFu;
            if (col >= 0) {
                const int64_t o = o0 + kept_before + __popc(km & ((1u << sub) - 1u));
                const T v = (T)__ddiv_rn((double)w, nrm);
                out_idx[o] = col;
                out_val[o] = v;
                s2 += (double)v * (double)v;
            }
            kept_before += __popc(km);
        }
        len = (uint32_t)kept_before;
#pragma unroll
        for (int d = 8; d > 0; d >>= 1) s2 += shfl_double(s2, lane ^ d);
        // an upper bound of the norm^2 of the row as stored, whatever the order of the additions
        n2 = __double2float_ru(s2 * (1.0 + 1e-12));
    }
    uint32_t nb = __float_as_uint(n2);   // non-negative floats order like unsigned integers
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        nb = max(nb, (uint32_t)__shfl_xor((int)nb, d, 64));
        len = max(len, (uint32_t)__shfl_xor((int)len, d, 64));
    }
    // one counter takes ~12 ns per atomic whoever sends it: look first (a stale value is lower, never higher)
    if (lane == 0 && props) {
        const volatile uint32_t *seen = props;
        if (nb > seen[1]) atomicMax(props + 1, nb);
        if (len > seen[2]) atomicMax(props + 2, len);
    }
}

// -------------------------------------------------------------------------------------------------
static void free_cache(sg_ctx *ctx, TokenCache &c) {
    ctx->release(c.d_longs);
    c.d_longs = nullptr;
    ctx->release(c.d_cnt);
    ctx->release(c.d_keys);
    ctx->release(c.d_tf);
    ctx->release(c.d_indptr);
    c = TokenCache();
}

static TokParams make_tok_params(const sg_vocab *v, const VocabImpl *im, const sg_strings *s) {
    TokParams tp;
    memset(&tp, 0, sizeof(tp));
    tp.ngram = v->params.ngram_size;
    tp.lower = (v->params.ascii_lower && !(s && s->prelowered)) ? 1 : 0;
    tp.bits = v->bits_per_char;
    for (int c = 0; c < 128; ++c)
        if (v->params.delete_table[c]) tp.del_mask[c >> 5] |= 1u << (c & 31);
    memcpy(tp.rank_of_byte, im->rank_of_byte, sizeof(tp.rank_of_byte));
    return tp;
}

// The strings of more than TOK_CAP n-grams that the wave-per-string kernels queued (c.d_longs): a workgroup per string, after
// one host round trip for their sizes.
template <typename KeyT, bool SYMBOLS>
static int tokenize_longs_t(sg_ctx *ctx, const sg_strings *s, const TokParams &tp, int32_t *df_table, int32_t df_replicas,
                            int64_t df_stride, TokenCache &c, uint32_t n_long) {
    if (n_long == 0) return SG_OK;
    uint32_t *longs = c.d_longs;
    KeyT *keys = (KeyT *)c.d_keys;
    // sizes of the long strings -> scratch offsets (characters: the length; keys: the next power of two)
    int64_t *d_sizes = nullptr, *d_coff = nullptr, *d_koff = nullptr;
    std::vector<int64_t> sizes(n_long), coff(n_long + 1, 0), koff(n_long + 1, 0);
    int st = sg_alloc(ctx, (size_t)n_long, &d_sizes);
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)n_long + 1, &d_coff);
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)n_long + 1, &d_koff);
    hipError_t e = hipSuccess;
    if (st == SG_OK) {
        hipLaunchKernelGGL(long_sizes_kernel, dim3((n_long + 255) / 256), dim3(256), 0, ctx->stream, s->d_offsets,
                           (const uint32_t *)(longs + 1), n_long, d_sizes);
        e = hipMemcpyAsync(sizes.data(), d_sizes, sizeof(int64_t) * n_long, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    }
    uint16_t *lchars = nullptr;
    KeyT *lkeys = nullptr;
    if (st == SG_OK && e == hipSuccess) {
        for (uint32_t i = 0; i < n_long; ++i) {
            int64_t p2 = 64;
            while (p2 < sizes[i]) p2 <<= 1;
            coff[i + 1] = coff[i] + sizes[i] + 8;
            koff[i + 1] = koff[i] + p2;
        }
        st = sg_alloc(ctx, (size_t)coff[n_long] + 8, &lchars);
        if (st == SG_OK) st = sg_alloc(ctx, (size_t)koff[n_long] + 8, &lkeys);
        if (st == SG_OK) {
            e = hipMemcpyAsync(d_coff, coff.data(), sizeof(int64_t) * (n_long + 1), hipMemcpyHostToDevice, ctx->stream);
            if (e == hipSuccess)
                e = hipMemcpyAsync(d_koff, koff.data(), sizeof(int64_t) * (n_long + 1), hipMemcpyHostToDevice, ctx->stream);
            if (e == hipSuccess) {
                hipLaunchKernelGGL((tokenize_long_kernel<KeyT, SYMBOLS>), dim3(n_long), dim3(256), 0, ctx->stream,
                                   (const void *)s->d_bytes, s->d_offsets, tp, (const uint32_t *)(longs + 1),
                                   (const int64_t *)d_coff, (const int64_t *)d_koff, lchars, lkeys,
                                   (const int64_t *)c.d_ub_ptr, c.d_cnt, keys, c.d_tf, df_table, df_replicas, df_stride);
                e = hipGetLastError();
            }
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);   // coff / koff are locals
        }
    }
    ctx->release(d_sizes);
    ctx->release(d_coff);
    ctx->release(d_koff);
    ctx->release(lchars);
    ctx->release(lkeys);
    if (st == SG_OK && e != hipSuccess) {
        sg_set_error("tokenize_long_kernel: %s", hipGetErrorString(e));
        st = SG_ERR_HIP;
    }
    return st;
}

// K1 over one string column: the one-wave-per-string kernel, then the workgroup-per-string kernel for the strings it
// handed over (one small host round trip: how many there are and how long they are).
// defer_longs (round 4): do NOT ask how many strings the last stage has to take -- the answer is "none" for any list of
// names, and asking is a synchronisation per column and fit.  The caller asks together with its own next question
// (sg_vec_fit_end: the size of the vocabulary) and tokenises the long strings then (resolve_longs), before anything
// that depends on them is final.
template <typename KeyT, bool SYMBOLS>
static int tokenize_set_t(sg_ctx *ctx, const sg_strings *s, const TokParams &tp, int32_t *df_table, int32_t df_replicas,
                          int64_t df_stride, TokenCache *out, bool defer_longs) {
    TokenCache c;
    c.src = s;
    c.n = s->n;
    uint32_t *mids = nullptr;    // [0] count, then the rows of more than 64 characters
    int st = sg_alloc(ctx, (size_t)s->n + 1, &c.d_cnt);
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)s->n + 2, &c.d_longs);
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)s->n + 2, &mids);
    c.d_ub_ptr = s->d_offsets;   // (no count, no scan: see TokenCache)
    // every row of L characters has at most L - n + 1 n-grams, so the total length bounds the padded size
    c.cap_total = s->total_bytes + 1;
    KeyT *keys = nullptr;
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)c.cap_total, &keys);
    c.d_keys = keys;
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)c.cap_total, &c.d_tf);
    uint32_t n_long = 0;
    if (st == SG_OK && s->n > 0) {
        // strings of up to 64 characters: tokenize_short_kernel; what it queues (mids): tokenize_kernel, launched on the
        // device-side count; what that one queues (longs, > TOK_CAP n-grams): tokenize_long_kernel after one host round trip
        uint32_t *longs = c.d_longs;
        hipError_t e = SG_ZERO2(ctx, longs, 4, mids, 4) == SG_OK ? hipSuccess : hipErrorUnknown;
        unsigned grid = (unsigned)ctx->num_cu * 8u;
        const int64_t wgs = (s->n + TOK_SHORT_WAVES - 1) / TOK_SHORT_WAVES;
        if ((int64_t)grid > wgs) grid = (unsigned)wgs;
        hipLaunchKernelGGL((tokenize_short_kernel<KeyT, SYMBOLS>), dim3(grid), dim3(64 * TOK_SHORT_WAVES), 0, ctx->stream,
                           (const void *)s->d_bytes, s->d_offsets, s->n, tp, (const int64_t *)c.d_ub_ptr, c.d_cnt, keys, c.d_tf,
                           df_table, df_replicas, df_stride, mids, mids + 1);
        grid = (unsigned)ctx->num_cu * 16u;
        if ((int64_t)grid > s->n) grid = (unsigned)s->n;
        hipLaunchKernelGGL((tokenize_kernel<KeyT, SYMBOLS>), dim3(grid), dim3(64), 0, ctx->stream, (const void *)s->d_bytes,
                           s->d_offsets, s->n, tp, (const int64_t *)c.d_ub_ptr, c.d_cnt, keys, c.d_tf, df_table, df_replicas,
                           df_stride, longs, longs + 1, (const uint32_t *)(mids + 1), (const uint32_t *)mids);
        if (e == hipSuccess) e = hipGetLastError();
        if (e == hipSuccess && !defer_longs) {
            e = hipMemcpyAsync(&n_long, longs, 4, hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        }
        if (e != hipSuccess) {
            sg_set_error("tokenize_kernel: %s", hipGetErrorString(e));
            st = SG_ERR_HIP;
        }
    }
    if (st == SG_OK && !defer_longs) {
        st = tokenize_longs_t<KeyT, SYMBOLS>(ctx, s, tp, df_table, df_replicas, df_stride, c, n_long);
        ctx->release(c.d_longs);
        c.d_longs = nullptr;
    }
    ctx->release(mids);
    if (st != SG_OK) {
        free_cache(ctx, c);
        return st;
    }
    *out = c;
    return SG_OK;
}

static int tokenize_set(sg_ctx *ctx, const sg_vocab *v, const sg_strings *s, int32_t *df_table, int32_t df_replicas,
                        int64_t df_stride, TokenCache *out, bool defer_longs = false) {
    const TokParams tp = make_tok_params(v, v->impl, s);
    const bool sym = s->sym_width == 2;
    if (v->sorted_mode)
        return sym ? tokenize_set_t<uint64_t, true>(ctx, s, tp, df_table, df_replicas, df_stride, out, defer_longs)
                   : tokenize_set_t<uint64_t, false>(ctx, s, tp, df_table, df_replicas, df_stride, out, defer_longs);
    return sym ? tokenize_set_t<uint32_t, true>(ctx, s, tp, df_table, df_replicas, df_stride, out, defer_longs)
               : tokenize_set_t<uint32_t, false>(ctx, s, tp, df_table, df_replicas, df_stride, out, defer_longs);
}

// the question tokenize_set_t(defer_longs) left open, answered: n_long strings of cache c go through the last stage
static int resolve_longs(sg_ctx *ctx, const sg_vocab *v, TokenCache &c, int32_t *df_table, int32_t df_replicas, int64_t df_stride,
                         uint32_t n_long) {
    const sg_strings *s = c.src;
    const TokParams tp = make_tok_params(v, v->impl, s);
    const bool sym = s->sym_width == 2;
    int st;
    if (v->sorted_mode)
        st = sym ? tokenize_longs_t<uint64_t, true>(ctx, s, tp, df_table, df_replicas, df_stride, c, n_long)
                 : tokenize_longs_t<uint64_t, false>(ctx, s, tp, df_table, df_replicas, df_stride, c, n_long);
    else
        st = sym ? tokenize_longs_t<uint32_t, true>(ctx, s, tp, df_table, df_replicas, df_stride, c, n_long)
                 : tokenize_longs_t<uint32_t, false>(ctx, s, tp, df_table, df_replicas, df_stride, c, n_long);
    ctx->release(c.d_longs);
    c.d_longs = nullptr;
    return st;
}

// fit = begin (tokenise, count document frequencies) + end (vocabulary).  The two halves are separate entry points so
// that a multi-GPU caller can all-reduce the dense document-frequency table in between (one rank tokenises one block of
// the strings; every rank then derives the SAME vocabulary and idf).
static int fit_begin(sg_ctx *ctx, const sg_strings *const *sets, int32_t n_sets, const sg_vec_params *params, bool df_marks,
                     sg_vocab **out) {
    SG_REQUIRE(ctx && sets && params && out && n_sets >= 1, "null argument");
    SG_REQUIRE(params->ngram_size >= 1 && params->ngram_size <= 64, "ngram_size must be in [1, 64]");
    SG_REQUIRE(params->dtype == SG_F32 || params->dtype == SG_F64, "dtype must be SG_F32 or SG_F64");
    for (int i = 0; i < n_sets; ++i) {
        SG_REQUIRE(sets[i] != nullptr, "null string set");
        SG_REQUIRE(sets[i]->sym_width == sets[0]->sym_width && sets[i]->alphabet == sets[0]->alphabet,
                   "the string columns of one fit must be all byte columns or all symbol columns of one alphabet");
    }
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
    const int n = params->ngram_size;
    // ---- character coding
    for (int c = 0; c < 128; ++c) {
        im->rank_of_byte[c] = (uint16_t)c;
        im->byte_of_rank[c] = (uint8_t)c;
    }
    v->bits_per_char = 7;
    if (sets[0]->sym_width == 2) {
        // symbol columns: the host ranked the characters into the alphabet of this fit (+ one code for "not in it")
        im->symbols = true;
        im->alphabet = sets[0]->alphabet;
        int bits = 1;
        while ((1 << bits) < im->alphabet) ++bits;
        v->bits_per_char = bits;
        im->local_alphabet = true;
    } else if (7 * n > 24) {
        // byte columns, long n-grams: ranks of the bytes that occur instead of the raw 7 bits
        bool lower_any = false;
        for (int i = 0; i < n_sets; ++i) lower_any |= params->ascii_lower && !sets[i]->prelowered;
        uint32_t *d_present = nullptr;
        st = sg_alloc(ctx, 4, &d_present);
        uint32_t present[4] = {0, 0, 0, 0};
        if (st == SG_OK) {
            (void)hipMemsetAsync(d_present, 0, 16, ctx->stream);
            for (int i = 0; i < n_sets; ++i)
                if (sets[i]->total_bytes > 0) {
                    const TokParams tp = make_tok_params(v, im, sets[i]);
                    hipLaunchKernelGGL(alphabet_kernel, dim3(1024), dim3(256), 0, ctx->stream, sets[i]->d_bytes,
                                       sets[i]->total_bytes, tp, d_present);
                }
            if (hipMemcpyAsync(present, d_present, 16, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                hipStreamSynchronize(ctx->stream) != hipSuccess)
                st = SG_ERR_HIP;
        }
        (void)lower_any;
        ctx->release(d_present);
        if (st == SG_OK) {
            // bytes that do not occur get the out-of-alphabet code: an n-gram containing one (only possible in a
            // later transform() of strings that were not part of fit()) is not in the vocabulary and is skipped
            for (int c = 0; c < 128; ++c) im->rank_of_byte[c] = SG_CHAR_ABSENT;
            int sigma = 0;
            for (int c = 0; c < 128; ++c)
                if ((present[c >> 5] >> (c & 31)) & 1u) {
                    im->rank_of_byte[c] = (uint16_t)sigma;
                    im->byte_of_rank[sigma] = (uint8_t)c;
                    ++sigma;
                }
            int bits = 1;
            while ((1 << bits) < sigma) ++bits;
            v->bits_per_char = bits;
            im->local_alphabet = true;
        }
    }
    if (st == SG_OK) {
        const int key_bits = v->bits_per_char * n;
        if (key_bits > 63) {
            sg_set_error("n-grams of %d characters over an alphabet of 2^%d need %d-bit keys; the device vocabulary holds 63",
                         n, v->bits_per_char, key_bits);
            st = SG_ERR_UNSUPPORTED;
        }
        v->sorted_mode = key_bits > 30;
        if (const char *e = ctx->opt("SG_VOCAB_SORTED"))      // test hook: the sorted vocabulary at any size
            if (e[0] == '1') v->sorted_mode = true;
        v->key_space = v->sorted_mode ? 0 : (int64_t)1 << key_bits;
    }
    if (st != SG_OK) {
        sg_vocab_free(v);
        return st;
    }

    {
        SgTimer timer(ctx, SG_K_TOKENIZE);
        int32_t replicas = 1;
        int64_t df_stride = 0;
        if (!v->sorted_mode) {
            // df table: `replicas` copies (row mod replicas picks one) while they stay small, summed afterwards
            df_stride = v->key_space + 1;
            replicas = 8;
            if (const char *e = ctx->opt("SG_DF_REPLICAS")) replicas = atoi(e);
            while (replicas > 1 && df_stride * replicas > ((int64_t)1 << 25)) replicas >>= 1;   // <= 128 MiB of counters
            if (replicas < 1) replicas = 1;
            if (const char *e = ctx->opt("SG_DF_MARKS")) df_marks = df_marks && e[0] != '0';   // A/B hook
            im->df_marks = df_marks;
            const int64_t copies = df_marks ? 1 : replicas;
            if (df_marks) replicas = 0;   // what the tokeniser kernels take as "mark, do not count"
            st = sg_alloc(ctx, (size_t)(df_stride * copies), &im->d_df_table);
            if (st == SG_OK) st = sg_alloc(ctx, (size_t)v->key_space + 1, &v->d_key_to_col);
            if (st == SG_OK)
                (void)hipMemsetAsync(im->d_df_table, 0, sizeof(int32_t) * (size_t)(df_stride * copies), ctx->stream);
        }
        for (int i = 0; i < n_sets && st == SG_OK; ++i) {
            TokenCache c;
            // (marks, dense table: the single-GPU fit, whose end asks for the long strings together with the vocabulary's size)
            st = tokenize_set(ctx, v, sets[i], im->d_df_table, replicas, df_stride, &c, !v->sorted_mode && df_marks);
            im->df_stride = df_stride;
            if (st == SG_OK) {
                im->caches.push_back(c);
                v->n_docs += sets[i]->n;
            }
        }
        if (st == SG_OK && !v->sorted_mode && replicas > 1) {
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

extern "C" int sg_vec_fit_begin(sg_ctx *ctx, const sg_strings *const *sets, int32_t n_sets, const sg_vec_params *params,
                                sg_vocab **out) {
    return fit_begin(ctx, sets, n_sets, params, /*df_marks=*/false, out);   // the caller may want the counters (sg_vocab_df_table)
}

extern "C" int sg_vocab_df_table(sg_vocab *v, int32_t **d_table, int64_t *n_entries, int32_t *shareable) {
    SG_REQUIRE(v && v->impl && d_table && n_entries, "null argument");
    SG_REQUIRE(v->n_terms == 0, "the vocabulary is already finished");
    SG_REQUIRE(!v->impl->df_marks, "this fit keeps marks, not counters");
    *d_table = v->impl->d_df_table;      // null in sorted mode
    *n_entries = v->key_space;
    // a table coded with the alphabet of the LOCAL strings (ngram_size > 3, symbol columns) means something else on
    // every rank; a sorted vocabulary has no table at all
    if (shareable) *shareable = (v->impl->local_alphabet || v->sorted_mode) ? 0 : 1;
    return SG_OK;
}

// sorted mode: vocabulary = sorted distinct keys of all columns of the fit, df = their run lengths
static int finish_sorted_vocabulary(sg_ctx *ctx, sg_vocab *v) {
    VocabImpl *im = v->impl;
    int64_t total = 0;
    std::vector<int64_t *> dst_ptrs;
    std::vector<int64_t> totals;
    int st = SG_OK;
    for (auto &c : im->caches) {   // where every row's keys go in the gathered array
        int64_t *d = nullptr;
        st = sg_alloc(ctx, (size_t)c.n + 1, &d);
        if (st != SG_OK) break;
        dst_ptrs.push_back(d);
        int64_t t = 0;
        if (c.n > 0) {
            st = sg_exclusive_scan_i32_to_i64(ctx, c.d_cnt, d, c.n);
            if (st == SG_OK && (hipMemcpyAsync(&t, d + c.n, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                                hipStreamSynchronize(ctx->stream) != hipSuccess))
                st = SG_ERR_HIP;
        }
        if (st != SG_OK) break;
        totals.push_back(t);
        total += t;
    }
    uint64_t *all = nullptr;
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)total + 1, &all);
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)total + 1, &v->d_keys);
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)total + 1, &v->d_df);
    int64_t at = 0;
    for (size_t i = 0; i < im->caches.size() && st == SG_OK; ++i) {
        const TokenCache &c = im->caches[i];
        if (c.n > 0 && totals[i] > 0) {
            hipLaunchKernelGGL(gather_keys_kernel, dim3((unsigned)((c.n + 255) / 256)), dim3(256), 0, ctx->stream,
                               (const int64_t *)c.d_ub_ptr, (const int32_t *)c.d_cnt, (const uint64_t *)c.d_keys,
                               (const int64_t *)dst_ptrs[i], c.n, all + at);
            if (hipGetLastError() != hipSuccess) st = SG_ERR_HIP;
        }
        at += totals[i];
    }
    int64_t n_unique = 0;
    if (st == SG_OK) st = sg_sort_unique_u64(ctx, all, total, v->d_keys, v->d_df, &n_unique);
    for (int64_t *d : dst_ptrs) ctx->release(d);
    ctx->release(all);
    if (st != SG_OK) return st;
    // an out-of-alphabet key cannot occur at fit (the alphabet comes from these very strings), but a symbol column may
    // carry the code on purpose: it sorts last and is not a term
    if (n_unique > 0) {
        uint64_t last = 0;
        if (hipMemcpyAsync(&last, v->d_keys + (n_unique - 1), 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess)
            return SG_ERR_HIP;
        if (last == SG_KEY_OOV64) --n_unique;
    }
    v->n_terms = n_unique;
    return SG_OK;
}

// marks form of the fit: df[col] = number of rows of all fitted columns that hold the column's n-gram
static int count_df_by_column(sg_ctx *ctx, sg_vocab *v) {
    VocabImpl *im = v->impl;
    const int32_t max_cols = 30 * 1024;   // 120 KiB of LDS counters per pass over the tokens
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void *)df_count_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, max_cols * 4);
        attr_done = true;
    }
    std::vector<int32_t> wgs;
    int32_t n_partial = 0;
    for (const auto &c : im->caches) {
        int64_t g = (c.n + 1023) / 1024;    // >= 1024 rows per workgroup, at most one workgroup per CU
        if (g > (int64_t)ctx->num_cu) g = (int64_t)ctx->num_cu;
        wgs.push_back((int32_t)g);
        n_partial += (int32_t)g;
    }
    uint32_t *partial = nullptr;
    int st = sg_alloc(ctx, (size_t)((int64_t)(n_partial > 0 ? n_partial : 1) * v->n_terms), &partial);
    if (st != SG_OK) return st;
    int32_t at = 0;
    const bool one_pass = v->n_terms <= max_cols && !(ctx->opt("SG_K2_COLUMNS") && ctx->opt("SG_K2_COLUMNS")[0] == '0');
    for (size_t i = 0; i < im->caches.size(); ++i) {
        TokenCache &c = im->caches[i];
        if (wgs[i] == 0) continue;
        for (int64_t col0 = 0; col0 < v->n_terms; col0 += max_cols) {
            const int32_t n_cols = (int32_t)(v->n_terms - col0 < max_cols ? v->n_terms - col0 : max_cols);
            hipLaunchKernelGGL(df_count_lds_kernel, dim3((unsigned)wgs[i]), dim3(1024), (size_t)n_cols * 4, ctx->stream,
                               (const int64_t *)c.d_ub_ptr, (const int32_t *)c.d_cnt, (uint32_t *)c.d_keys,
                               (const int32_t *)v->d_key_to_col, c.n, (int32_t)col0, n_cols, v->n_terms,
                               partial + (int64_t)at * v->n_terms, one_pass ? 1 : 0);
        }
        if (one_pass) c.keys_are_columns = true;   // (one pass saw every token: its keys are columns now)
        at += wgs[i];
    }
    hipLaunchKernelGGL(df_sum_kernel, dim3((unsigned)((v->n_terms + 63) / 64)), dim3(256), 0, ctx->stream, (const uint32_t *)partial,
                       n_partial, v->n_terms, v->d_df);
    if (hipGetLastError() != hipSuccess) {
        sg_set_error("df_count_lds_kernel: %s", hipGetErrorString(hipGetLastError()));
        st = SG_ERR_HIP;
    }
    ctx->release(partial);
    return st;
}

extern "C" int sg_vec_fit_end(sg_ctx *ctx, sg_vocab *v, int64_t n_docs_total) {
    SG_REQUIRE(ctx && v && v->impl, "null argument");
    SG_REQUIRE(v->n_terms == 0, "the vocabulary is already finished");
    VocabImpl *im = v->impl;
    if (n_docs_total > 0) v->n_docs = n_docs_total;
    int st = SG_OK;
    {
        SgTimer timer(ctx, SG_K_VOCAB);
        if (v->sorted_mode) {
            st = finish_sorted_vocabulary(ctx, v);
        } else {
            // ---- vocabulary = keys with df > 0, column id = rank
            uint32_t *d_total = nullptr;
            st = sg_alloc(ctx, 4, &d_total);
            if (st == SG_OK) {
                const unsigned grid = (unsigned)((v->key_space + 255) / 256);
                (void)grid;   // (the scan itself asks "does the key occur": no pass that writes 0 / 1 first)
                st = sg_exclusive_scan_positive_i32(ctx, im->d_df_table, (uint32_t *)v->d_key_to_col, v->key_space, d_total);
            }
            uint32_t n_terms = 0;
            for (int attempt = 0; attempt < 2 && st == SG_OK; ++attempt) {
                // the size of the vocabulary and -- the question fit_begin left open -- whether any column holds strings for the
                // last tokeniser stage: ONE synchronisation; the row pointers of the columns' matrices ride along (TokenCache)
                std::vector<uint32_t> n_long(im->caches.size(), 0u);
                std::vector<int64_t> nnz_of(im->caches.size(), -1);
                for (size_t q = 0; q < im->caches.size() && st == SG_OK; ++q) {
                    TokenCache &c = im->caches[q];
                    if (c.n <= 0) continue;
                    if (!c.d_indptr) st = sg_alloc(ctx, (size_t)c.n + 1, &c.d_indptr);
                    if (st == SG_OK) st = sg_exclusive_scan_i32_to_i64(ctx, c.d_cnt, c.d_indptr, c.n);
                }
                if (st != SG_OK) break;
                // (into PINNED memory: a copy to a pageable address is staged and waited for on the spot -- three copies were three
                //  round trips, 20 - 40 us of idle GPU each; layout: [0] vocabulary, then per column {strings for the last stage,
                //  -, non-zeros: 64 bits})
                uint32_t *hf = ctx->h_fetch;
                const bool pinned = 4 * im->caches.size() + 4 <= SG_H_FETCH_WORDS;
                hipError_t e = hipMemcpyAsync(pinned ? (void *)hf : (void *)&n_terms, d_total, 4, hipMemcpyDeviceToHost, ctx->stream);
                for (size_t q = 0; q < im->caches.size() && e == hipSuccess; ++q) {
                    if (im->caches[q].d_longs)
                        e = hipMemcpyAsync(pinned ? (void *)(hf + 2 + 4 * q) : (void *)&n_long[q], im->caches[q].d_longs, 4, hipMemcpyDeviceToHost, ctx->stream);
                    if (e == hipSuccess && im->caches[q].d_indptr)
                        e = hipMemcpyAsync(pinned ? (void *)(hf + 4 + 4 * q) : (void *)&nnz_of[q], im->caches[q].d_indptr + im->caches[q].n, 8,
                                           hipMemcpyDeviceToHost, ctx->stream);
                }
                if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
                if (pinned && e == hipSuccess) {
                    n_terms = hf[0];
                    for (size_t q = 0; q < im->caches.size(); ++q) {
                        if (im->caches[q].d_longs) n_long[q] = hf[2 + 4 * q];
                        if (im->caches[q].d_indptr) memcpy(&nnz_of[q], hf + 4 + 4 * q, 8);
                    }
                }
                for (size_t q = 0; q < im->caches.size(); ++q) im->caches[q].nnz = nnz_of[q];
                if (e != hipSuccess) {
                    sg_set_error("reading the vocabulary size failed: %s", hipGetErrorString(e));
                    st = SG_ERR_HIP;
                    break;
                }
                bool any_long = false;
                for (size_t q = 0; q < im->caches.size() && st == SG_OK; ++q) {
                    if (!im->caches[q].d_longs) continue;
                    any_long = any_long || n_long[q] > 0;
                    st = resolve_longs(ctx, v, im->caches[q], im->d_df_table, 0, im->df_stride, n_long[q]);   // (marks: replicas 0)
                }
                if (!any_long || st != SG_OK) break;
                // long strings have marked keys of their own: the vocabulary is taken again (and their rows' counts were not
                // final when the row pointers were summed: the next attempt sums them again)
                const unsigned grid = (unsigned)((v->key_space + 255) / 256);
                (void)grid;
                st = sg_exclusive_scan_positive_i32(ctx, im->d_df_table, (uint32_t *)v->d_key_to_col, v->key_space, d_total);
            }
            ctx->release(d_total);
            if (st == SG_OK) {
                v->n_terms = n_terms;
                if (v->n_terms > 0) {
                    st = sg_alloc(ctx, (size_t)v->n_terms + 1, &v->d_keys);
                    if (st == SG_OK) st = sg_alloc(ctx, (size_t)v->n_terms + 1, &v->d_df);
                    if (st == SG_OK) {
                        const unsigned grid = (unsigned)((v->key_space + 255) / 256);
                        hipLaunchKernelGGL(vocab_finalize_kernel, dim3(grid), dim3(256), 0, ctx->stream, im->d_df_table,
                                           v->key_space, v->d_key_to_col, v->d_keys, v->d_df);
                        if (hipGetLastError() != hipSuccess) st = SG_ERR_HIP;
                    }
                    if (st == SG_OK && im->df_marks) st = count_df_by_column(ctx, v);
                }
            }
        }
        if (st == SG_OK && v->n_terms == 0) {
            sg_set_error("empty vocabulary; perhaps the documents only contain stop words");
            st = SG_ERR_BADARG;
        }
    }
    return st;
}

extern "C" int sg_vec_fit(sg_ctx *ctx, const sg_strings *const *sets, int32_t n_sets, const sg_vec_params *params,
                          sg_vocab **out) {
    SG_REQUIRE(out != nullptr, "null argument");
    sg_vocab *v = nullptr;
    SG_TRY(fit_begin(ctx, sets, n_sets, params, /*df_marks=*/true, &v));
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

extern "C" int sg_vocab_coding(const sg_vocab *v, int32_t *bits_per_char, int32_t *symbols, int32_t *sorted_mode) {
    SG_REQUIRE(v && v->impl, "vocab is null");
    if (bits_per_char) *bits_per_char = v->bits_per_char;
    if (symbols) *symbols = v->impl->symbols ? 1 : 0;
    if (sorted_mode) *sorted_mode = v->sorted_mode ? 1 : 0;
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
    for (int64_t i = 0; i < v->n_terms; ++i) df[i] = hdf[(size_t)i];
    return SG_OK;
}

extern "C" int sg_vocab_byte_alphabet(const sg_vocab *v, uint8_t *byte_of_code, int32_t *n_codes) {
    SG_REQUIRE(v && v->impl && byte_of_code && n_codes, "null argument");
    SG_REQUIRE(!v->impl->symbols, "the vocabulary was fitted on symbol columns: the caller holds their alphabet");
    *n_codes = 1 << v->bits_per_char;
    if (*n_codes > 128) *n_codes = 128;
    memcpy(byte_of_code, v->impl->byte_of_rank, (size_t)*n_codes);
    return SG_OK;
}

// Weights a caller supplies: positive and finite?  (The rows K2 makes are "cosine-like by construction" only then:
// non-negative entries, norm 1 -- sg_csr_props; a zero, negative or NaN weight sends the matrix through the measured path.)
static bool weights_positive_and_finite(const void *w, int64_t n, int32_t dtype) {
    if (dtype == SG_F64) {
        const double *p = (const double *)w;
        for (int64_t i = 0; i < n; ++i)
            if (!(p[i] > 0.0) || !(p[i] <= 1.79e308)) return false;
    } else {
        const float *p = (const float *)w;
        for (int64_t i = 0; i < n; ++i)
            if (!(p[i] > 0.f) || !(p[i] <= 3.4e38f)) return false;
    }
    return true;
}

extern "C" int sg_vocab_set_idf(sg_ctx *ctx, sg_vocab *v, const void *idf, int32_t dtype) {
    SG_REQUIRE(ctx && v && idf, "null argument");
    SG_REQUIRE(dtype == v->params.dtype, "idf dtype differs from the vectoriser dtype");
    const size_t s = dtype == SG_F64 ? 8 : 4;
    v->idf_trusted = weights_positive_and_finite(idf, v->n_terms, dtype);
    if (!v->d_idf) SG_TRY(ctx->alloc(((size_t)v->n_terms + 1) * s, &v->d_idf));
    SG_HIP_TRY(hipMemcpyAsync(v->d_idf, idf, s * (size_t)v->n_terms, hipMemcpyHostToDevice, ctx->stream));
    SG_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SG_OK;
}

template <typename T>
__global__ void __launch_bounds__(256) idf_from_table_kernel(const int32_t *__restrict__ df, int64_t n_terms, const T *__restrict__ table,
                                                             int64_t n_docs, T *__restrict__ idf) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_terms) return;
    int64_t d = df[k];
    if (d < 0) d = 0;
    if (d > n_docs) d = n_docs;
    idf[k] = table[d];
}

extern "C" int sg_ctx_put_idf_table(sg_ctx *ctx, int64_t n_docs, int32_t dtype, const void *table) {
    SG_REQUIRE(ctx && table && n_docs >= 0, "null argument");
    SG_REQUIRE(dtype == SG_F32 || dtype == SG_F64, "dtype must be SG_F32 or SG_F64");
    const size_t bytes = ((size_t)n_docs + 1) * (dtype == SG_F64 ? 8 : 4);
    void *d = nullptr;
    SG_HIP_TRY(hipMalloc(&d, bytes));
    if (hipMemcpyAsync(d, table, bytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess) {
        (void)hipFree(d);
        sg_set_error("upload of the idf table failed");
        return SG_ERR_HIP;
    }
    // (entry 0 -- a term no document holds -- is never gathered for a column of the vocabulary: df >= 1 there)
    const bool trusted = weights_positive_and_finite((const char *)table + (dtype == SG_F64 ? 8 : 4), n_docs, dtype);
    for (auto &t : ctx->idf_tables)
        if (t.n_docs == n_docs && t.dtype == dtype) {      // replaced (scans in flight are behind the synchronisation above)
            (void)hipFree(t.d);
            t.d = d;
            t.trusted = trusted;
            return SG_OK;
        }
    if (ctx->idf_tables.size() >= 4) {                      // the oldest goes
        (void)hipFree(ctx->idf_tables.front().d);
        ctx->idf_tables.erase(ctx->idf_tables.begin());
    }
    ctx->idf_tables.push_back({n_docs, dtype, d, trusted});
    return SG_OK;
}

extern "C" int sg_vocab_apply_idf_table(sg_ctx *ctx, sg_vocab *v, int32_t *applied) {
    SG_REQUIRE(ctx && v && applied, "null argument");
    *applied = 0;
    SG_REQUIRE(v->d_df != nullptr || v->n_terms == 0, "the vocabulary has no document counts (fit first)");
    const void *table = nullptr;
    bool trusted = false;
    for (const auto &t : ctx->idf_tables)
        if (t.n_docs == v->n_docs && t.dtype == v->params.dtype) {
            table = t.d;
            trusted = t.trusted;
        }
    if (!table) return SG_OK;
    v->idf_trusted = trusted;
    const size_t s = v->params.dtype == SG_F64 ? 8 : 4;
    if (!v->d_idf) SG_TRY(ctx->alloc(((size_t)v->n_terms + 1) * s, &v->d_idf));
    if (v->n_terms > 0) {
        const unsigned grid = (unsigned)((v->n_terms + 255) / 256);
        if (v->params.dtype == SG_F64)
            hipLaunchKernelGGL(idf_from_table_kernel<double>, dim3(grid), dim3(256), 0, ctx->stream, (const int32_t *)v->d_df, v->n_terms,
                               (const double *)table, v->n_docs, (double *)v->d_idf);
        else
            hipLaunchKernelGGL(idf_from_table_kernel<float>, dim3(grid), dim3(256), 0, ctx->stream, (const int32_t *)v->d_df, v->n_terms,
                               (const float *)table, v->n_docs, (float *)v->d_idf);
        SG_HIP_TRY(hipGetLastError());
    }
    *applied = 1;
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
        delete im;
    }
    ctx->release(v->d_key_to_col);
    ctx->release(v->d_keys);
    ctx->release(v->d_df);
    ctx->release(v->d_idf);
    delete v;
    return SG_OK;
}

template <typename T, typename KeyT, typename Lookup>
static void launch_weight(sg_ctx *ctx, const TokenCache *tc, Lookup lookup, const sg_vocab *v, int64_t n, const int64_t *indptr,
                          int32_t *idx, void *val, uint32_t *props) {
    const char *pe = ctx->opt("SG_K2_PLAIN");   // A/B and test hook: the thread-per-row statement of the arithmetic
    const bool plain = pe && pe[0] == '1';
    if (plain)
        hipLaunchKernelGGL((weight_normalize_kernel<T, KeyT, Lookup>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                           (const int64_t *)tc->d_ub_ptr, (const int32_t *)tc->d_cnt, (const KeyT *)tc->d_keys,
                           (const int32_t *)tc->d_tf, lookup, (const T *)v->d_idf, n, indptr, idx, (T *)val, props);
    else
        hipLaunchKernelGGL((weight_normalize_rows16_kernel<T, KeyT, Lookup>), dim3((unsigned)((n + 15) / 16)), dim3(256), 0,
                           ctx->stream, (const int64_t *)tc->d_ub_ptr, (const int32_t *)tc->d_cnt, (const KeyT *)tc->d_keys,
                           (const int32_t *)tc->d_tf, lookup, (const T *)v->d_idf, n, indptr, idx, (T *)val, props);
}

extern "C" int sg_vec_transform(sg_ctx *ctx, const sg_vocab *v, const sg_strings *strings, sg_csr **out) {
    SG_REQUIRE(ctx && v && strings && out, "null argument");
    SG_REQUIRE(v->d_idf != nullptr, "sg_vocab_set_idf has not been called");
    VocabImpl *im = impl_of(v);
    SG_REQUIRE(im != nullptr, "unknown vocab");
    SG_REQUIRE((strings->sym_width == 2) == im->symbols && (!im->symbols || strings->alphabet == im->alphabet),
               "the strings are not of the kind (bytes / symbols of this alphabet) the vocabulary was fitted on");
    // tokens: reuse the pass made by fit() when these strings were part of it
    TokenCache local;
    const TokenCache *tc = nullptr;
    for (const auto &c : im->caches)
        if (c.src == strings && c.n == strings->n) tc = &c;
    int st = SG_OK;
    if (!tc) {
        SgTimer timer(ctx, SG_K_TOKENIZE);
        st = tokenize_set(ctx, v, strings, nullptr, 1, 0, &local);
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
    int64_t nnz = 0;
    const DenseLookup dense{v->d_key_to_col};
    const SortedLookup sorted{v->d_keys, v->n_terms};
    {
        SgTimer timer(ctx, SG_K_WEIGHT);
        // the tokens of a column that was part of fit() are all in the vocabulary (min_df = 1): kept == distinct n-grams
        const bool all_kept = tc != &local;
        TokenCache *fitted = all_kept ? const_cast<TokenCache *>(tc) : nullptr;
        const bool have_ptr = fitted && fitted->d_indptr && fitted->nnz >= 0 && n > 0;   // (summed and read by the fit's end)
        if (have_ptr) {
            indptr = fitted->d_indptr;
            nnz = fitted->nnz;
            fitted->d_indptr = nullptr;      // the matrix owns them now; a second transform of the column sums again
            fitted->nnz = -1;
        }
        if (!all_kept) st = sg_alloc(ctx, (size_t)n + 1, &kept);
        if (st == SG_OK && !have_ptr) st = sg_alloc(ctx, (size_t)n + 1, &indptr);
        if (have_ptr) {
            ;
        } else if (st == SG_OK && n > 0) {
            const unsigned grid = (unsigned)((n + 255) / 256);
            if (all_kept)
                ;
            else if (v->sorted_mode)
                hipLaunchKernelGGL((kept_count_kernel<uint64_t, SortedLookup>), dim3(grid), dim3(256), 0, ctx->stream,
                                   (const int64_t *)tc->d_ub_ptr, (const int32_t *)tc->d_cnt, (const uint64_t *)tc->d_keys,
                                   sorted, n, kept);
            else
                hipLaunchKernelGGL((kept_count_kernel<uint32_t, DenseLookup>), dim3(grid), dim3(256), 0, ctx->stream,
                                   (const int64_t *)tc->d_ub_ptr, (const int32_t *)tc->d_cnt, (const uint32_t *)tc->d_keys,
                                   dense, n, kept);
            st = sg_exclusive_scan_i32_to_i64(ctx, all_kept ? tc->d_cnt : kept, indptr, n);
        } else if (st == SG_OK) {
            (void)hipMemsetAsync(indptr, 0, sizeof(int64_t), ctx->stream);
        }
        if (st == SG_OK && !have_ptr) {
            if (hipMemcpyAsync(&nnz, indptr + n, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                hipStreamSynchronize(ctx->stream) != hipSuccess)
                st = SG_ERR_HIP;
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
        m->from_vectoriser = v->idf_trusted;
        // (K2 can leave the largest row norm and the longest row behind; nothing reads them unless the opt-in row blocks
        //  are built -- the matrix is cosine-like by construction, sg_csr_props)
        if (st == SG_OK && ctx->opt("SG_ROW_BLOCKS") && ctx->opt("SG_ROW_BLOCKS")[0] == '1') {
            st = sg_alloc(ctx, (size_t)4, &m->d_props_words);
            if (st == SG_OK && hipMemsetAsync(m->d_props_words, 0, 16, ctx->stream) != hipSuccess) st = SG_ERR_HIP;
        }
        if (st == SG_OK && n > 0) {
            uint32_t *props = m->d_props_words;
            if (v->sorted_mode) {
                if (m->dtype == SG_F64) launch_weight<double, uint64_t>(ctx, tc, sorted, v, n, indptr, idx, val, props);
                else launch_weight<float, uint64_t>(ctx, tc, sorted, v, n, indptr, idx, val, props);
            } else if (tc->keys_are_columns) {   // (the fit's df pass left the columns in place of the keys)
                const ColumnsLookup cols{};
                if (m->dtype == SG_F64) launch_weight<double, uint32_t>(ctx, tc, cols, v, n, indptr, idx, val, props);
                else launch_weight<float, uint32_t>(ctx, tc, cols, v, n, indptr, idx, val, props);
            } else {
                if (m->dtype == SG_F64) launch_weight<double, uint32_t>(ctx, tc, dense, v, n, indptr, idx, val, props);
                else launch_weight<float, uint32_t>(ctx, tc, dense, v, n, indptr, idx, val, props);
            }
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
