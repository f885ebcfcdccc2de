// Internal declarations shared by the translation units of libsg_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <map>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/sg_hip.h"

#define SG_WAVE 64
#define SG_NUM_XCD 8

void sg_set_error(const char *fmt, ...);

#define SG_HIP_TRY(expr)                                                                        \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess) {                                                                 \
            sg_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return _e == hipErrorOutOfMemory ? SG_ERR_OOM : SG_ERR_HIP;                         \
        }                                                                                       \
    } while (0)

#define SG_TRY(expr)                  \
    do {                              \
        int _s = (expr);              \
        if (_s != SG_OK) return _s;   \
    } while (0)

#define SG_REQUIRE(cond, msg)                   \
    do {                                        \
        if (!(cond)) {                          \
            sg_set_error("bad argument: %s", msg); \
            return SG_ERR_BADARG;               \
        }                                       \
    } while (0)

// ---------------------------------------------------------------------------------------------
// Context: device, stream, a caching scratch pool (so that a repeated "step" allocates nothing),
// per-kernel HIP events.
#define SG_H_FETCH_WORDS 2048
struct sg_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int num_cu = 256;
    size_t lds_per_cu = 160 * 1024;
    size_t total_mem = 0;                // device memory, bytes (hipDeviceProp_t::totalGlobalMem)

    std::mutex mu;
    std::mutex scan_mu;                          // the single-pass scan's bookkeeping (epoch, ticket base, descriptor array)
    std::multimap<size_t, void *> free_blocks;   // size -> ptr
    std::map<void *, size_t> live_blocks;        // ptr -> size

    hipEvent_t ev_start[SG_K_COUNT];
    hipEvent_t ev_stop[SG_K_COUNT];
    bool ev_valid[SG_K_COUNT];
    int64_t spgemm_entry_bytes = 0;              // 4 + s of the most recent multiply
    int64_t spgemm_fixed_bytes = 0;              // its algorithmic bytes that do not scale with MACs
    double prune_row_bytes = 0.0;                // pruned multiply: bytes read per pair scored exactly (row pointers + mean packed row)
    double prune_q8_bytes = 0.0;                 // ... and per candidate of the first filter checked by the second (mean 8-bit copy; 0: no second filter)
    bool prune_symmetric = false;                // the most recent multiply took the self-join form of the pruned kernel
    double pilot_ms_pruned = 0.0, pilot_ms_exact = 0.0;   // the most recent pruned-or-exact pilot's two estimates (0: none ran)
    int64_t *d_stat_words = nullptr;             // [0]=macs [1]=out_nnz [2..4]=pruned rows/postings/survivors [5]=rows handed to K4 [6]=pairs scored exactly
    int64_t *h_stat_words = nullptr;             // pinned mirror
    uint32_t *h_fetch = nullptr;                 // pinned, SG_H_FETCH_WORDS words: where small read-backs land (a copy to pageable memory is a round trip of its own, one per copy)

    // Tuning switches (SG_*): read from the environment ONCE, when the context is created, changed only through
    // sg_ctx_set_option, listed by sg_ctx_options -- nothing inside an API call looks at the environment (a variable left
    // in a user's shell must not silently change what a later call does: SG_PRUNE=0 is a 23 x slower multiply).
    std::map<std::string, std::string> opts;
    const char *opt(const char *name) const {    // value of a switch, or null: not set
        auto it = opts.find(name);
        return it == opts.end() ? nullptr : it->second.c_str();
    }
    bool poison = false;                         // SG_POISON_ALLOC=1 (copied out of opts: read under the pool's lock)

    // single-pass prefix sums (sg_api.hip: scan_lookback_kernel): tile descriptors + the ticket counter behind them,
    // owned by the context and reused by every scan (scans of one context run one after the other on its stream)
    unsigned long long *d_scan_desc = nullptr;
    size_t scan_desc_cap = 0;
    uint32_t scan_ticket_base = 0, scan_epoch = 0;
    struct IdfTable {                            // sg_ctx_put_idf_table: idf as a function of the document count
        int64_t n_docs;
        int32_t dtype;
        void *d;
        bool trusted;                            // every weight positive and finite (checked on the host copy)
    };
    std::vector<IdfTable> idf_tables;
    int inner_multiply_depth = 0;                // > 0: sg_spgemm_topn runs for a wrapper (groups of identical rows) that counts the kept entries itself
    bool group_table_overflow = false;           // sg_collapse.hip: the table path met a group too large for it

    int alloc(size_t bytes, void **out);         // pooled hipMalloc
    void release(void *p);                       // back to the pool
    void trim();
};

template <typename T>
static inline int sg_alloc(sg_ctx *ctx, size_t count, T **out) {
    void *p = nullptr;
    int s = ctx->alloc(count * sizeof(T), &p);
    *out = (T *)p;
    return s;
}

struct SgTimer {   // records start/stop events of one kernel group on the context stream
    sg_ctx *ctx;
    int which;
    SgTimer(sg_ctx *c, int w) : ctx(c), which(w) { (void)hipEventRecord(ctx->ev_start[w], ctx->stream); }
    ~SgTimer() {
        (void)hipEventRecord(ctx->ev_stop[which], ctx->stream);
        ctx->ev_valid[which] = true;
    }
};

// ---------------------------------------------------------------------------------------------
struct sg_strings {
    sg_ctx *ctx = nullptr;
    const uint8_t *d_bytes = nullptr;
    const int64_t *d_offsets = nullptr;
    int64_t n = 0;
    int64_t total_bytes = 0;             // bytes, or symbols of a symbol column
    bool owned = false;
    // kind of column (string_grouper_amd/strprep.py): UTF-8 bytes the tokeniser filters itself, or uint16 symbols that
    // the host has already lower-cased, regex-deleted and ranked into the fit's alphabet (0xFFFF = not in it)
    int32_t sym_width = 1;               // 1: d_bytes are bytes; 2: d_bytes are uint16 symbols, offsets count symbols
    int32_t alphabet = 0;                // symbol columns: number of symbols of the alphabet
    bool prelowered = false;             // byte columns: the host applied str.lower() already, leave A-Z alone
};

struct sg_csr {
    sg_ctx *ctx = nullptr;
    int64_t n_rows = 0, n_cols = 0, nnz = 0;
    int32_t dtype = SG_F32;
    const int64_t *d_indptr = nullptr;   // absolute offsets into d_indices / d_data (a row-block view
    const int32_t *d_indices = nullptr;  // keeps the parent's arrays and a shifted d_indptr pointer)
    const void *d_data = nullptr;
    bool owned = false;
    // lazily computed by sg_csr_props (sg_spgemm_pruned.hip): 0 unknown, 1 cosine-like, 2 not
    mutable int props_state = 0;
    mutable float props_max_norm2 = 0.f;
    mutable uint32_t props_max_nnz = 0;  // longest row
    bool from_vectoriser = false;        // made by sg_vec_transform (K2): cosine-like by construction (sg_csr_props)
    mutable bool props_by_construction = false;   // the two above are the vectoriser's guarantees, not measurements
    // a matrix made by the vectoriser is cosine-like by construction; K2 leaves [0] violations (= 0), [1] max ||row||^2 as
    // float bits, [2] longest row here and sg_csr_props reads them instead of scanning the matrix again (owned)
    uint32_t *d_props_words = nullptr;
    // groups of identical rows of a LEFT matrix (sg_spgemm_topn, one-sided products: round 4), made on first use and kept
    // for the multiplies that follow with the same matrix (column blocks of the right-hand side): 0 not looked at yet,
    // 1 looked at and not worth it / not possible, 2 `left_groups` holds them (owned, also by views)
    mutable int left_state = 0;
    mutable struct SgCollapse *left_groups = nullptr;
    struct SgCollapse *rows_of = nullptr;   // this matrix is `unique` of these groups: its rows may be pending (sg_csr_ensure_rows)
};

struct SgScoreCtx {
    const uint32_t *fwd_ptr = nullptr;   // packed rows of B, a uint2 per position q: {first entry, the row's own index};
                                         // row at q = entries [fwd_ptr[2 q], fwd_ptr[2 q + 2])
    const void *fwd = nullptr;
    const uint32_t *orig_of = nullptr;   // position -> row; null: identity
    // Row blocks (round 3): the same rows at a FIXED stride of blk_bytes (a multiple of 128), 128-byte aligned: entry 0 is
    // a header {the row's own index, its number of entries}, entries 1 .. nnz follow ({term, value}: 8 bytes f32 / 16
    // bytes f64), zero padding up to the end of the last 128-byte line the row uses (lines behind it are never read).
    // Scoring a pair then needs no row pointer first, its lines are aligned, and the line behind the first is requested
    // as soon as the header is there.  null: no blocks (a row would need more than 1 KiB): the packed rows above are used.
    const void *blk = nullptr;
    uint32_t blk_bytes = 0;
    // Second filter (round 5): an 8-bit copy of every right-hand row at a FIXED stride of SG_Q8_STRIDE bytes (two 128-byte
    // lines), 256-byte aligned, no pointer to fetch first:
    //   word 0  first entry of the row among the packed rows (`fwd`)        } what the exact scoring needs of the row: a
    //   word 1  the row's own index (position -> row)                       } candidate that passes costs no pointer fetch
    //   word 2  the row's entries (bits [0, 31)); bit 31: no 8-bit copy (more than SG_Q8_MAX_ENTRIES entries: always passes)
    //   word 3  0
    //   words 4 .. 63  entries in ascending term order, (term << 8) | bq, bq = ceil(value / norm_up * 255) -- rounded UP, so
    //           sum_k a_k * bq_k * norm_up / 255 is an upper bound of the pair's score; zero behind the row's last entry
    //           (only the 16-byte units a row uses are written, and read: 28 entries fit the first 128-byte line)
    // A candidate of the first filter whose bound stays under the threshold is not scored (sg_spgemm_pruned.hip,
    // drain_survivors).  q8_scale = 255 / norm_up, rounded down.  null: no second filter (terms beyond 24 bits, or switched
    // off: SG_Q8=0).
    const uint4 *q8 = nullptr;
    float q8_scale = 0.f;
    uint32_t pad_ = 0;
};
#define SG_Q8_STRIDE 256u
#define SG_Q8_MAX_ENTRIES 60u   // 16 units of 16 bytes: the header, then four entries a unit

// Groups of identical right-hand rows (sg_collapse.hip): the index is built over one representative per group.
struct SgCollapse {
    sg_ctx *ctx = nullptr;
    int64_t n_orig = 0, n_u = 0;         // rows of the caller's matrix, groups
    uint32_t *d_gid = nullptr;           // n_orig: group of every row (groups numbered by ascending representative)
    uint32_t *d_group_ptr = nullptr;     // n_u + 1
    uint32_t *d_members = nullptr;       // n_orig: rows of group g = members[group_ptr[g] .. group_ptr[g + 1]), ascending
    uint32_t *d_rep_rows = nullptr;      // n_u: lowest row of every group
    int64_t *d_rep_start = nullptr;      // (table path) n_u: where the representative's row starts in the caller's arrays
    int32_t *d_rep_len = nullptr;        //              n_u: and its entries -- one load level for whoever copies the rows
    struct sg_csr *unique = nullptr;     // the representatives' rows (owned)
    // round 4: the rows of `unique` may not have been WRITTEN yet (row pointers and sizes are final): they are the rows
    // d_rep_rows of pending_src, and the index build writes them together with its own copies of them (one read of the
    // source instead of three passes: sg_postings.hip, gather_rows_kernel); sg_collapse_materialize writes them alone
    // round 6: NOBODY on the way to the self-join's index reads them, so the index build no longer writes them either
    // (88 MB at 663 k): whoever does read the representatives' matrix asks sg_csr_ensure_rows first.  With d_rep_len set the
    // row POINTERS of `unique` are pending as well.
    const struct sg_csr *pending_src = nullptr;
};

struct sg_postings {
    sg_ctx *ctx = nullptr;
    int64_t n_right = 0, n_terms = 0, nnz = 0;
    int32_t dtype = SG_F32;
    int32_t tile_log2 = 12;
    int32_t n_tiles = 0;
    // seg[k * n_tiles + t] .. seg[k * n_tiles + t + 1] = postings of term k whose row lies in tile t
    // the CSR the postings were built from (borrowed: it must outlive the postings); the pruned multiply
    // scores its surviving candidates by merging row i of A with row j of this matrix
    const int64_t *b_indptr = nullptr;
    const int32_t *b_indices = nullptr;
    const void *b_data = nullptr;
    uint32_t *d_seg = nullptr;           // n_terms * n_tiles + 1
    uint32_t *d_term_len = nullptr;      // n_terms: entries of term k's list (= seg[(k+1) * n_tiles] - seg[k * n_tiles])
    uint32_t *d_term_start = nullptr;    // n_terms + 1: start of term k's list (LDS build path; sg_postings_ensure_full re-reads it)
    int32_t *d_rows = nullptr;           // nnz   (row j of B)                  } null until sg_postings_ensure_full when the
    void *d_vals = nullptr;              // nnz   (value B[j, k])               } build left the postings proper out
    sg_csr src;                          // the matrix the index was built over (rows in position order; not owned)
    int32_t split = 1;                   // parts a tile was counted in (sg_postings.hip, LDS path)
    // rows of B packed for the pruned multiply's exact scoring (built only for cosine-like B):
    // f32: {int32 term, float value} (8 B), f64: {int32 term, pad, double value} (16 B); row j = entries
    // [d_fwd_ptr[j], d_fwd_ptr[j+1])
    void *d_fwd = nullptr;               // (null when the row blocks below are built: they replace it)
    uint32_t *d_fwd_ptr = nullptr;       // (n_right + 1) x {pointer, the row's own index (position -> row)}
    void *d_blk = nullptr;               // row blocks at a fixed stride (SgScoreCtx::blk)
    uint32_t blk_bytes = 0;
    void *d_q8 = nullptr;                // 8-bit copies of the rows at a fixed stride (SgScoreCtx::q8), (n_right + 1) records
    // 4-byte "filter postings", same order as the postings proper (only for cosine-like B): the column inside its tile
    // (or super-tile), the value and the norm of the row's frequent part (terms of list length >= freq_min), both
    // quantised UPWARDS relative to norm_up.  The fields differ between the tile-by-tile form (fold_log2 == 0) and the
    // stream form: sg_postings.hip, emit_posting, has both layouts and what the multiply's instructions read from them.
    uint32_t *d_filt = nullptr;
    // Position space (sg_postings.hip, "column permutation").  The multiply's column tiles are ranges of consecutive
    // right-hand rows; on a SORTED list similar names are neighbours, a row's candidates pile up in a few tiles and the
    // pruned multiply runs 2.6 x slower than on the same names shuffled.  The index is therefore built over a fixed
    // permutation of the rows: position p holds row orig_of[p] (pos_of is the inverse), `permuted` is the matrix with its
    // rows in position order (owned; the self-join form multiplies IT), and the kernels turn a position back into a row
    // wherever an index leaves them: the result's row, the columns they keep (ties are broken by the ORIGINAL column),
    // the pairs they exchange.  All null: positions are rows.
    sg_csr *permuted = nullptr;
    uint32_t *d_orig_of = nullptr, *d_pos_of = nullptr;
    // what the pruned multiply's survivor routine needs of the index, as ONE struct in device memory: that routine is a
    // real call inside the tile loop, and every argument it takes is a register the loop cannot use at the call sites
    struct SgScoreCtx *d_score_ctx = nullptr;   // [0] with the second filter (when built), [1] without
    // per term a 16-byte aligned row of nt_pad entries: d_ends[k * nt_pad + t] = BYTE offset into d_filt of the end
    // of segment (k, t) (entries past the last tile repeat the end of the list): one 16-byte load = four tiles
    uint32_t *d_ends = nullptr;
    int32_t nt_pad = 0;
    // Stream form of the pruned multiply: 2^fold_log2 consecutive tiles (a super-tile, one visit of a left row) share one
    // accumulator tile; the filter postings then carry the tile index mod 2^fold_log2 and a narrower bq
    // (sg_postings.hip, emit_posting).  d_ends8[k * nv_pad + v] = byte offset into d_filt of the end of term k's postings
    // in super-tiles 0 .. v (16-byte aligned rows, the tail repeats the end of the list, one all-zero row behind the last
    // term).  fold_log2 == 0: the tile-by-tile form and its posting format.
    int32_t fold_log2 = 0;
    uint32_t *d_ends8 = nullptr;
    int32_t nv_pad = 0;
    float norm_up = 0.f;                 // max ||row of B||, rounded up
    uint32_t freq_min = 0;               // list length from which a term counts as frequent
    bool cosine_like = false;            // B: values >= 0, sorted rows, row norms <= 1 (sg_csr_props)
    float max_norm2 = 0.f;               // max ||row of B||^2, rounded up
    // Identical rows collapsed (sg_collapse.hip): everything above indexes the representatives' matrix (n_right = groups);
    // the caller's matrix is (caller_*), the multiply expands its result to the caller's rows / columns.  `plain`: an index
    // over all rows, built on demand for what the collapsed one does not serve (top_n > 64, the multi-GPU row ranges).
    SgCollapse *collapse = nullptr;
    const sg_csr *caller_b = nullptr;    // (borrowed: outlives the postings, like b_indptr)
    sg_csr caller_b_copy;                // the struct itself, copied (the caller may free its handle's wrapper, not the arrays)
    int64_t n_right_caller = 0;
    mutable sg_postings *plain = nullptr;
    // The exact kernel's own layout of the same rows (its tile: twice the waves per CU of the pruned multiply's 4096 columns;
    // postings proper only), built on demand when the exact kernel runs the whole product in the self-join form --
    // thresholds below the pruned kernel's envelope: 14.6 -> 12.1 ms at 200 k names and 0.4 (scripts/exact_sym_probe.py).
    mutable sg_postings *exact_native = nullptr;
    // The pruned multiply's TILE-BY-TILE form of the same rows (2048-column tiles, an accumulator per column), built on demand
    // for self-joins at thresholds below SG_ALT_FORM_BELOW (0.65): the stream form folds eight tiles onto one accumulator tile
    // and its false alarms grow as the threshold falls -- 200 k names at 0.5: 9.6 ms against 5.6 (scripts/form_sweep.py).
    mutable sg_postings *alt_tile = nullptr;
    sg_csr built_from;                   // the matrix handed to the build (a copy of the struct, arrays borrowed: the caller keeps it alive -- include/sg_hip.h)
    bool built_from_valid = false;
    bool tile_form = false;              // this index IS such a one (SG_POSTINGS_TILE_FORM)
    const sg_postings *view_of = nullptr;   // a shallow copy (the index seen without its groups): the object that owns what is built on demand
    int32_t build_tile_cols = 0, build_flags = 0;
};

struct sg_topn {
    sg_ctx *ctx = nullptr;
    int64_t n_rows = 0, n_cols = 0;
    int32_t stride = 0;
    int32_t dtype = SG_F32;
    int32_t top_n_asked = 0;             // a result over GROUPS of identical rows: the caller's top_n (the stride is cut at the
                                         // number of groups, the expansion to member rows must not be -- sg_topn_expand_groups)
    int32_t *d_cols = nullptr;
    void *d_vals = nullptr;
    int32_t *d_counts = nullptr;
};

struct sg_vocab {
    sg_ctx *ctx = nullptr;
    sg_vec_params params;
    int32_t bits_per_char = 7;
    int64_t n_terms = 0, n_docs = 0;
    int64_t key_space = 0;               // dense mode: 2^(bits_per_char * ngram_size)
    // dense mode (bits_per_char * ngram_size <= 30): 32-bit keys, a table over the whole key space;
    // sorted mode (up to 64 bits): 64-bit keys, the vocabulary is the sorted array d_keys, columns by binary search
    bool sorted_mode = false;
    int32_t *d_key_to_col = nullptr;     // dense mode: key_space entries, -1 = not in vocabulary
    uint64_t *d_keys = nullptr;          // n_terms: key of column i (ascending)
    int32_t *d_df = nullptr;             // n_terms
    void *d_idf = nullptr;               // n_terms, params.dtype; null until sg_vocab_set_idf
    // the weights are positive and finite (checked where they arrive from the caller: sg_vocab_set_idf,
    // sg_ctx_put_idf_table): only then are the rows K2 makes cosine-like BY CONSTRUCTION (sg_csr_props); otherwise the
    // matrix is measured like any caller's matrix before the pruned multiply may rely on its bounds
    bool idf_trusted = false;
    struct VocabImpl *impl = nullptr;    // tokeniser state (sg_vectorize.hip): token caches, character coding, df table
};

// sg_matchlist.hip
int sg_matchlist_device_view(const sg_matchlist *ml, int64_t *n_rows, int64_t *n_cols, int64_t *n_entries, int32_t *dtype,
                             const int64_t **row_ptr, const int32_t **cols, const void **vals);

#define SG_POSTINGS_TILE_FORM (1 << 11)    // sg_postings_build_flags (internal): the pruned multiply's tile-by-tile form, 2048-column tiles, no 8-bit rows
#define SG_POSTINGS_EXACT_ONLY (1 << 10)   // sg_postings_build_flags (internal): no filter postings, packed rows, 8-bit rows
// sg_spgemm_pruned.hip
int sg_csr_props(sg_ctx *ctx, const sg_csr *m, bool *cosine_like, float *max_norm2, uint32_t *max_nnz = nullptr);
bool sg_pruned_supports_tile(int32_t tile_log2);
int sg_postings_ensure_full(sg_ctx *ctx, const sg_postings *p);
bool sg_q8_applies(const sg_ctx *ctx, const sg_postings *Bt, double threshold);   // sg_spgemm_pruned.hip: second filter in this call?   // sg_postings.hip: the exact kernel's postings, on demand
int sg_spgemm_pruned_launch(sg_ctx *ctx, const sg_csr *A, const sg_postings *Bt, int32_t keep, sg_topn *r,
                            double threshold, double delta, uint32_t *row_counter,
                            uint32_t *flagged_count, uint32_t *flagged_rows, unsigned long long *stats);

int sg_spgemm_pruned_symmetric(sg_ctx *ctx, const sg_csr *A, const sg_postings *Bt, int32_t keep, sg_topn *r,
                               double threshold, double delta, unsigned long long *stats, bool *done, int64_t row_lo = 0,
                               int64_t row_hi = -1 /* the whole matrix */, int32_t **export_pairs = nullptr,
                               int64_t *export_n = nullptr, int64_t row_step = 1 /* rows row_hi - 1, row_hi - 1 - step, ... */,
                               bool exact_all = false /* every row through the exact kernel's self-join launch (thresholds below the pruned kernel's) */);
int sg_selfjoin_merge_pairs(sg_ctx *ctx, sg_topn *r, const int32_t *d_pairs, int64_t n_pairs, int64_t row_lo, int64_t row_hi,
                            const uint32_t *pos_of /* row -> position when the range is one of positions; null: rows */,
                            int64_t row_step = 1);

// The pair list of the self-join form: the MIRRORED pairs (i, j < i, score) above the threshold, in chunks of
// SG_PAIR_CHUNK entries that a wave owns while it fills them (one returning atomic per chunk).  Written by the pruned
// kernel and -- for the rows that kernel cannot take -- by the exact kernel's self-join launch (sg_spgemm_topn.hip).
#define SG_PAIR_CHUNK 256u
#define SG_PAIR_NO_CHUNK 0xFFFFFFFFu
struct SgPairSink {
    uint32_t *d_i = nullptr;
    uint32_t *d_j = nullptr;
    void *d_s = nullptr;
    uint32_t *d_row_count = nullptr;           // mirrored matches per row (counted as the pairs are written)
    uint32_t *d_chunk_count = nullptr;         // entries used of every chunk
    uint32_t *d_chunks_used = nullptr;         // chunks handed out
    unsigned long long *d_totals = nullptr;    // pairs in closed chunks
    uint32_t chunks = 0;
};
// sg_spgemm_topn.hip: the rows of `row_list` (device-side length) through the exact kernel in its self-join form --
// a row keeps its matches j <= i and appends the mirrored pairs to the sink, exactly as the pruned kernel's rows do.
unsigned sg_spgemm_exact_selfjoin_grid(const sg_ctx *ctx, const sg_postings *Bt = nullptr /* given: the grid of a launch over ALL rows */);
int sg_spgemm_exact_selfjoin_rows(sg_ctx *ctx, const sg_csr *A, const sg_postings *Bt, int32_t keep, sg_topn *r,
                                  double threshold, uint32_t *row_counter, const uint32_t *row_list,
                                  const uint32_t *row_list_len, const SgPairSink &sink, bool all_rows = false);

// sg_collapse.hip
int sg_collapse_build(sg_ctx *ctx, const sg_csr *B, SgCollapse **out, bool left_side = false, bool defer_rows = false);
int sg_collapse_materialize(sg_ctx *ctx, SgCollapse *c);
int sg_csr_ensure_rows(sg_ctx *ctx, const sg_csr *m);
void sg_collapse_free(SgCollapse *c);
int sg_collapse_expand(sg_ctx *ctx, const SgCollapse *c, const sg_topn *ru, bool rows_are_groups, sg_topn *out,
                       const int32_t *row_list = nullptr);   // row_list: output row k is the caller's row row_list[k]
// sg_sortvocab.hip: stable sort of (key, value) pairs by key
int sg_sort_pairs_u64_u32(sg_ctx *ctx, const uint64_t *d_keys, const uint32_t *d_vals, int64_t n, uint64_t *d_keys_out,
                          uint32_t *d_vals_out);
// sg_sortvocab.hip: ascending distinct values of d_keys[0 .. n) and how often each occurs (d_keys is overwritten)
int sg_sort_unique_u64(sg_ctx *ctx, uint64_t *d_keys, int64_t n, uint64_t *d_unique, int32_t *d_counts, int64_t *n_unique);

// exclusive prefix sum of n uint32 values (in place allowed); total written to *d_total if non-null
// several arrays cleared by ONE launch (bytes and pointers multiples of four)
int sg_zero_ranges(sg_ctx *ctx, int n, void *const *ptrs, const size_t *bytes);
#define SG_ZERO2(ctx, p0, b0, p1, b1) [&]() { void *const pp_[] = {(void *)(p0), (void *)(p1)}; const size_t bb_[] = {(size_t)(b0), (size_t)(b1)}; return sg_zero_ranges(ctx, 2, pp_, bb_); }()
#define SG_ZERO3(ctx, p0, b0, p1, b1, p2, b2) [&]() { void *const pp_[] = {(void *)(p0), (void *)(p1), (void *)(p2)}; const size_t bb_[] = {(size_t)(b0), (size_t)(b1), (size_t)(b2)}; return sg_zero_ranges(ctx, 3, pp_, bb_); }()
#define SG_ZERO4(ctx, p0, b0, p1, b1, p2, b2, p3, b3) [&]() { void *const pp_[] = {(void *)(p0), (void *)(p1), (void *)(p2), (void *)(p3)}; const size_t bb_[] = {(size_t)(b0), (size_t)(b1), (size_t)(b2), (size_t)(b3)}; return sg_zero_ranges(ctx, 4, pp_, bb_); }()
// d_out[i] = number of entries > 0 before i (d_out may not alias d_in: the types differ in meaning, not in size)
int sg_exclusive_scan_positive_i32(sg_ctx *ctx, const int32_t *d_in, uint32_t *d_out, int64_t n, uint32_t *d_total);
int sg_exclusive_scan_u32(sg_ctx *ctx, const uint32_t *d_in, uint32_t *d_out, int64_t n, uint32_t *d_total);
// same for int64 outputs from int32 inputs (row pointers)
int sg_exclusive_scan_i32_to_i64(sg_ctx *ctx, const int32_t *d_in, int64_t *d_out, int64_t n);
