/* sg_hip.h -- C ABI of libsg_hip.so, the MI355X (gfx950) core for string_grouper's hot path.
 *
 * Boundary.  Every entry point replaces a call the reference makes
 * into a third-party native dependency from string_grouper/string_grouper.py (paths relative
 * to the reference tree):
 *
 *   seam b1  TfidfVectorizer(min_df=1, analyzer=n_grams, dtype).fit / .transform
 *            string_grouper.py:306 (construct), :699-707 (fit), :689-692 (transform)
 *            -> sg_strings_*, sg_vec_fit, sg_vocab_*, sg_vec_transform
 *   seam b2  sparse_dot_topn.sp_matmul_topn     string_grouper.py:725-732, :737-743
 *            -> sg_postings_build (the B.transpose() + CSC->CSR step) + sg_spgemm_topn,
 *               or the one-shot host mirror sg_sp_matmul_topn_host
 *            sparse_dot_topn.zip_sp_matmul_topn string_grouper.py:746
 *            -> sg_topn_zip
 *
 * Conventions
 *   - plain C, no torch / scipy types.  Pointers named d_* are DEVICE pointers (HBM), all other
 *     pointers are host pointers.  A caller that already owns device memory (e.g. a torch tensor's
 *     data_ptr()) uses the *_from_device constructors; nothing is copied and nothing is freed.
 *   - every function returns an sg_status; sg_last_error() gives a thread-local message.
 *     SG_ERR_OVERFLOW is what the Python layer turns into OverflowError, the one exception the
 *     reference handles around _build_matches (string_grouper.py:397-413).
 *   - calls are asynchronous on the context's HIP stream unless they return data to the host.
 *   - value type: SG_F32 or SG_F64 (the only two sparse_dot_topn accepts, string_grouper.py:18).
 *   - column/row indices are int32, row pointers int64.
 *   - result order ("sort" != 0): per row by (score descending, column ascending); ties at the
 *     top-n cut are resolved by that same order; values must be STRICTLY greater than threshold.
 */
#ifndef SG_HIP_H
#define SG_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    SG_OK = 0,
    SG_ERR_BADARG = 1,
    SG_ERR_OOM = 2,       /* hipMalloc failed            -> MemoryError   */
    SG_ERR_OVERFLOW = 3,  /* index space exhausted       -> OverflowError */
    SG_ERR_HIP = 4,       /* any other HIP runtime error -> RuntimeError  */
    SG_ERR_NODEVICE = 5,  /* no gfx950 device visible    -> RuntimeError  */
    SG_ERR_UNSUPPORTED = 6
} sg_status;

enum { SG_F32 = 0, SG_F64 = 1 };

typedef struct sg_ctx sg_ctx;           /* one per (process, GPU): device, stream, scratch pool  */
typedef struct sg_strings sg_strings;   /* device-resident string column: bytes + int64 offsets  */
typedef struct sg_vocab sg_vocab;       /* fitted vocabulary: n-gram key -> column, df, idf       */
typedef struct sg_csr sg_csr;           /* device-resident CSR matrix                             */
typedef struct sg_postings sg_postings; /* device-resident inverted index of B (= B^T), bucketed
                                           by column tile for the LDS accumulator of the multiply */
typedef struct sg_topn sg_topn;         /* device-resident fixed-stride top-n result             */

/* ------------------------------------------------------------------ library / context */
const char *sg_last_error(void);
/* Bumped whenever a signature or a struct of this header changes (round 4: 2 -- row_step arguments of round 3, sg_stats
 * grew; round 5: 3 -- sg_stats.prune_scored); a binding compares it with the value it was written for right after loading
 * the library. */
#define SG_ABI_VERSION 3
int sg_abi_version(void);
int sg_device_count(int *count);
/* hip_stream: a hipStream_t to launch on (e.g. torch.cuda.current_stream().cuda_stream), or NULL
 * to let the library create its own non-blocking stream. */
int sg_ctx_create(int device, void *hip_stream, sg_ctx **out);
int sg_ctx_destroy(sg_ctx *ctx);
int sg_ctx_sync(sg_ctx *ctx);
/* Release cached scratch back to the driver. */
int sg_ctx_trim(sg_ctx *ctx);
/* Tuning switches (SG_PRUNE, SG_SYM, SG_PRUNE_DELTA ... -- README, "tuning knobs").  The library reads the SG_* variables
 * of the environment ONCE, in sg_ctx_create; afterwards only these calls change them (value NULL: unset), and nothing
 * inside an API call looks at the environment.  sg_ctx_options writes the active set as NAME=VALUE lines into buf and
 * returns the bytes needed (0-terminated); sg_ctx_reset_options re-reads the environment (test hook).  Not to be
 * called while another thread has a call in flight on the same context. */
int sg_ctx_set_option(sg_ctx *ctx, const char *name, const char *value);
int sg_ctx_reset_options(sg_ctx *ctx);
int sg_ctx_options(sg_ctx *ctx, char *buf, int64_t len);

/* ------------------------------------------------------------------ strings (input of seam b1) */
/* Arrow large_string layout: string i = bytes[offsets[i] .. offsets[i+1]). */
int sg_strings_from_host(sg_ctx *ctx, const uint8_t *bytes, const int64_t *offsets, int64_t n,
                         sg_strings **out);
int sg_strings_from_device(sg_ctx *ctx, const uint8_t *d_bytes, const int64_t *d_offsets, int64_t n,
                           int64_t total_bytes, sg_strings **out);
/* A column whose characters the host has already lower-cased, regex-deleted and ranked into the alphabet of the fit
 * (n-grams over non-ASCII code points: normalize_to_ascii=False, string_grouper.py:202, :373-374): symbols[i] is the
 * rank of a character in the sorted alphabet of alphabet_size characters, 0xFFFF = "not in the alphabet" (only in a
 * later transform); offsets count symbols.  All columns of one fit must share the alphabet. */
int sg_strings_from_host_symbols(sg_ctx *ctx, const uint16_t *symbols, const int64_t *offsets, int64_t n,
                                 int32_t alphabet_size, sg_strings **out);
/* Byte column: the host has already applied str.lower() (rows with non-ASCII characters go through the host, whose
 * NFKD step can produce upper-case ASCII that must stay: 'TM' from U+2122); the device then leaves A-Z alone. */
int sg_strings_set_prelowered(sg_strings *s, int32_t prelowered);
int sg_strings_free(sg_strings *s);

/* ------------------------------------------------------------------ seam b1: vectoriser */
typedef struct {
    int32_t ngram_size;         /* StringGrouperConfig.ngram_size (string_grouper.py:189)         */
    int32_t ascii_lower;        /* != 0: map 'A'-'Z' to 'a'-'z' on the device (ignore_case)       */
    int32_t dtype;              /* SG_F32 / SG_F64  (tfidf_matrix_dtype)                          */
    int32_t reserved;
    uint8_t delete_table[128];  /* != 0: ASCII byte removed before n-gramming (the regex, :376)   */
} sg_vec_params;

/* TfidfVectorizer.fit(concat(sets)): learns the vocabulary (column = rank of the n-gram in
 * code-point order, exactly sklearn's sorted vocabulary) and the document frequencies.
 * Bytes >= 0x80 are dropped (== .encode('ascii','ignore') after the host's NFKD).
 * Returns SG_ERR_BADARG with "empty vocabulary" if no n-gram exists at all. */
int sg_vec_fit(sg_ctx *ctx, const sg_strings *const *sets, int32_t n_sets, const sg_vec_params *params,
               sg_vocab **out);
/* The two halves of sg_vec_fit, for a caller that shards the strings over several GPUs: every rank tokenises ITS
 * block (begin), the dense document-frequency tables are summed across ranks (sg_vocab_df_table gives the device
 * pointer: key_space int32 counters, e.g. for an RCCL all-reduce), and every rank derives the same vocabulary
 * from the summed table (end; n_docs_total = documents over all ranks, it enters the idf).
 * *shareable == 0: the keys are coded with the alphabet of the local strings (ngram_size > 3) and the table
 * must not be combined with another rank's -- fit on the whole column instead. */
int sg_vec_fit_begin(sg_ctx *ctx, const sg_strings *const *sets, int32_t n_sets, const sg_vec_params *params,
                     sg_vocab **out);
int sg_vocab_df_table(sg_vocab *v, int32_t **d_table, int64_t *n_entries, int32_t *shareable);
int sg_vec_fit_end(sg_ctx *ctx, sg_vocab *v, int64_t n_docs_total);
int sg_vocab_size(const sg_vocab *v, int64_t *n_terms, int64_t *n_docs);
/* How the keys of this vocabulary are coded: bits per character; != 0 if fitted on symbol columns; != 0 if the
 * vocabulary is the sorted key array (keys wider than 30 bits) instead of the dense table. */
int sg_vocab_coding(const sg_vocab *v, int32_t *bits_per_char, int32_t *symbols, int32_t *sorted_mode);
/* keys[i]: the n-gram of column i, its character codes packed big-endian bits_per_char bits each (sg_vocab_coding);
 * a code is the rank of a symbol in the caller's alphabet (symbol columns) or indexes sg_vocab_byte_alphabet (byte
 * columns: the identity for n-grams of up to 3 characters).  df[i]: its document count. */
int sg_vocab_to_host(sg_ctx *ctx, const sg_vocab *v, uint64_t *keys, int64_t *df);
int sg_vocab_byte_alphabet(const sg_vocab *v, uint8_t *byte_of_code /* 128 */, int32_t *n_codes);
/* idf is computed by the caller from df (numpy, sklearn's exact op sequence text.py:1664-1679,
 * so that log() is bit-identical) and installed here; dtype must match params.dtype. */
int sg_vocab_set_idf(sg_ctx *ctx, sg_vocab *v, const void *idf, int32_t dtype);
/* The same without the round trip (round 4).  idf of a term depends on its document count and the number of documents
 * only -- idf = f(df; n_docs) -- so the caller hands the library, ONCE per (n_docs, dtype), the table f(0 .. n_docs) made
 * with its own numpy (n_docs + 1 values; sg_ctx_put_idf_table, kept by the context, a handful of them), and every later fit
 * over as many documents weights its terms on the device: idf[column] = table[df[column]] (sg_vocab_apply_idf_table;
 * *applied == 0: the context has no table for this vocabulary's n_docs and dtype -- put one, or use sg_vocab_set_idf).
 * A fit() then needs neither sg_vocab_to_host nor an upload; keys and counts still come to the host when asked for. */
int sg_ctx_put_idf_table(sg_ctx *ctx, int64_t n_docs, int32_t dtype, const void *table);
int sg_vocab_apply_idf_table(sg_ctx *ctx, sg_vocab *v, int32_t *applied);
int sg_vocab_free(sg_vocab *v);
/* TfidfVectorizer.transform(strings): counts -> *idf -> row L2 normalise, CSR with sorted indices. */
int sg_vec_transform(sg_ctx *ctx, const sg_vocab *v, const sg_strings *strings, sg_csr **out);

/* ------------------------------------------------------------------ CSR objects */
int sg_csr_from_host(sg_ctx *ctx, int64_t n_rows, int64_t n_cols, const int64_t *indptr,
                     const int32_t *indices, const void *data, int32_t dtype, sg_csr **out);
/* (A wrapped matrix is read where it lies.  The library keeps what it derives from a matrix WITH the sg_csr object --
 *  its properties, and from 65 536 rows the groups of identical rows a one-sided sg_spgemm_topn multiplies once -- so the
 *  arrays must not be rewritten while the object lives: wrap the new contents in a new object instead, as for an index.) */
int sg_csr_from_device(sg_ctx *ctx, int64_t n_rows, int64_t n_cols, int64_t nnz, const int64_t *d_indptr,
                       const int32_t *d_indices, const void *d_data, int32_t dtype, sg_csr **out);
int sg_csr_dims(const sg_csr *m, int64_t *n_rows, int64_t *n_cols, int64_t *nnz, int32_t *dtype);
int sg_csr_device_ptrs(const sg_csr *m, const int64_t **d_indptr, const int32_t **d_indices,
                       const void **d_data);
int sg_csr_to_host(sg_ctx *ctx, const sg_csr *m, int64_t *indptr, int32_t *indices, void *data);
/* Rows [r0, r1) as a view (no copy); the parent must outlive the view.  (A view derives its own groups of identical
 * rows when it is multiplied; it never shares the parent's.) */
int sg_csr_row_block(sg_ctx *ctx, const sg_csr *m, int64_t r0, int64_t r1, sg_csr **out);
int sg_csr_free(sg_csr *m);

/* Row-wise similarity of two matrices of the same shape (StringGrouper.dot / compute_pairwise_similarities,
 * string_grouper.py:433-440: np.asarray(master_matrix.multiply(duplicate_matrix).sum(axis=1))):
 * out_host[i] = sum over the common columns of A[i, k] * B[i, k], products rounded to the value type and
 * added in the order scipy + numpy use (ascending column; first + pairwise sum of the rest), so the result
 * is bit-identical.  out_host: n_rows values of the matrices' type.  Rows must be sorted by column. */
int sg_csr_rowwise_dot(sg_ctx *ctx, const sg_csr *A, const sg_csr *B, void *out_host);

/* ------------------------------------------------------------------ seam b2: sparse top-n multiply */
/* Inverted index of B (n_right x V): for every term k the (row j, value) pairs, grouped by column
 * tile j / tile_cols.  tile_cols must be a power of two supported by the multiply (0 = default).
 * The postings keep a reference to B's arrays (the multiply re-scores candidates against B's rows, and an index over
 * all rows is built from them on demand when a multiply asks for more than 64 columns per row of an index over groups of
 * identical rows): B MUST stay alive -- and unchanged -- until the postings are freed. */
int sg_postings_build(sg_ctx *ctx, const sg_csr *B, int32_t tile_cols, sg_postings **out);
/* The same with options.  SG_POSTINGS_NO_PERMUTATION: by default the index is built over a fixed permutation of B's rows
 * (a sorted name list has its similar names side by side, which piles a row's candidates into a few column tiles: the
 * pruned multiply ran 2.6 x slower on 663 k sorted names than on the same names shuffled); results never show it --
 * rows, columns and the order of equal scores are B's own.  (The ranges of the multi-GPU self-join form,
 * sg_selfjoin_range, are ranges of positions then: sg_postings_permutation.) */
enum { SG_POSTINGS_NO_PERMUTATION = 1 };
int sg_postings_build_flags(sg_ctx *ctx, const sg_csr *B, int32_t tile_cols, int32_t flags, sg_postings **out);
int sg_postings_free(sg_postings *p);

/* C = topn_rowwise(A . B^T restricted to > threshold).  A: n_left x V, postings of B: n_right x V.
 * Row i of the result holds counts[i] <= top_n entries at [i*top_n, i*top_n + counts[i]). */
int sg_spgemm_topn(sg_ctx *ctx, const sg_csr *A, const sg_postings *Bt, int32_t top_n, double threshold,
                   int32_t sort, sg_topn **out);
int sg_topn_dims(const sg_topn *r, int64_t *n_rows, int32_t *stride, int32_t *dtype, int64_t *n_cols);
int sg_topn_device_ptrs(const sg_topn *r, const int32_t **d_cols, const void **d_vals, const int32_t **d_counts);
int sg_topn_to_host(sg_ctx *ctx, const sg_topn *r, int32_t *cols, void *vals, int32_t *counts);
int sg_topn_counts_to_host(sg_ctx *ctx, const sg_topn *r, int32_t *counts);   /* the per-row counts only */
/* Upload a fixed-stride result held on the host (used to feed host CSR blocks to sg_topn_zip). */
int sg_topn_from_host(sg_ctx *ctx, int64_t n_rows, int64_t n_cols, int32_t stride, int32_t dtype,
                      const int32_t *cols, const void *vals, const int32_t *counts, sg_topn **out);
/* The same from DEVICE memory (a device-to-device copy on the context's stream): the multi-GPU path hands its all-gathered
 * blocks over without a trip through the host (string_grouper.py:750 vstack, on the device). */
int sg_topn_from_device(sg_ctx *ctx, int64_t n_rows, int64_t n_cols, int32_t stride, int32_t dtype,
                        const int32_t *d_cols, const void *d_vals, const int32_t *d_counts, sg_topn **out);
/* zip_sp_matmul_topn: parts[b] = A . B_b^T; columns of part b are offset by col_offsets[b]. */
int sg_topn_zip(sg_ctx *ctx, const sg_topn *const *parts, const int64_t *col_offsets, int32_t n_parts,
                int32_t top_n, sg_topn **out);
int sg_topn_free(sg_topn *r);

/* One-shot host mirror of sparse_dot_topn.sp_matmul_topn(A, B.T, top_n, threshold, sort):
 * uploads A and B (both CSR over the same V columns), multiplies, downloads.  out_* hold
 * n_left * top_n (cols, vals) and n_left counts. */
int sg_sp_matmul_topn_host(sg_ctx *ctx, int64_t n_left, int64_t n_right, int64_t n_cols,
                           const int64_t *a_indptr, const int32_t *a_indices, const void *a_data,
                           const int64_t *b_indptr, const int32_t *b_indices, const void *b_data,
                           int32_t dtype, int32_t top_n, double threshold, int32_t sort,
                           int32_t *out_cols, void *out_vals, int32_t *out_counts);

/* ------------------------------------------------------------------ match list (fit() tail) */
/* Device version of what fit() does with the multiply's result (string_grouper.py:417-431, :755-763):
 * fix_diagonal: every (r, r) := 1 (added when absent, :954-958); symmetrize: every stored (r, c) also
 * stored as (c, r) (:960-964); rows come back sorted by column.  With both flags 0 it is a plain
 * fixed-stride -> CSR compaction that keeps the within-row order unless sort_by_column is set (the
 * reference's vstack(..., dtype=np.float64) at :750 re-sorts the rows of a float32 result by column as a
 * side effect of scipy's up-cast; a float64 result keeps the multiply's order).  Entry i of row r is
 * (r, cols[row_ptr[r] + i], vals[...]). */
typedef struct sg_matchlist sg_matchlist;
int sg_matchlist_build(sg_ctx *ctx, const sg_topn *r, int32_t fix_diagonal, int32_t symmetrize,
                       int32_t sort_by_column, sg_matchlist **out);
int sg_matchlist_dims(const sg_matchlist *ml, int64_t *n_rows, int64_t *n_entries, int32_t *dtype);
int sg_matchlist_to_host(sg_ctx *ctx, const sg_matchlist *ml, int64_t *row_ptr, int32_t *cols, void *vals);
int sg_matchlist_free(sg_matchlist *ml);

/* The two reductions the reference runs over the match list, on the device: one int32 per string comes
 * back instead of the list.
 * sg_matchlist_best_master (match_most_similar, string_grouper.py:803-807: groupby('dupe_side') max
 *   similarity, then min master_side): out_best[c] = the row with the largest similarity in column c, the
 *   lowest such row among equals, -1 when the column has no entry.  out_best: n_cols of the list (host).
 * sg_matchlist_group_reps (group_similar_strings, string_grouper.py:851-904: scipy connected_components
 *   + group_rep): out_rep[i] = the representative of string i's group -- centroid == 0: the member with
 *   the lowest index ('first'); centroid != 0: the member with the largest row sum of similarities, the
 *   lowest index among equals.  Needs a square list (self-join).  out_rep: n_rows (host). */
int sg_matchlist_best_master(sg_ctx *ctx, const sg_matchlist *ml, int32_t *out_best);
int sg_matchlist_group_reps(sg_ctx *ctx, const sg_matchlist *ml, int32_t centroid, int32_t *out_rep);

/* Cost estimate of every left row for load balancing across GPUs: out_cost[i] = number of
 * intermediate products row i generates = sum over its non-zeros of the posting-list length.
 * (The reference has no analogue: its n_blocks[0] split is by row count, string_grouper.py:714-722.) */
int sg_row_costs(sg_ctx *ctx, const sg_csr *A, const sg_postings *Bt, int64_t *out_cost);

/* Self-join across GPUs.  One GPU scores every pair (i, j) of a self-join once, from the row
 * with the larger index (half the work of the one-sided multiply the reference's n_blocks[0] split implies,
 * string_grouper.py:714-752).  Across ranks that form splits by LEFT-ROW RANGES of the same, replicated matrix:
 *
 * sg_selfjoin_range: the rows [row_lo, row_hi) of A against the columns j <= i.  A must be the matrix Bt was built
 *   from.  *out: a result object over ALL rows of A in which the rows of the range hold their matches j <= i (the
 *   other rows are empty); *d_pairs: the mirrored pairs -- that row i of the range matches row j < i, which may belong
 *   to another rank -- as *n_pairs records of *pair_words int32 words: {i, j, score bits} (f32) or {i, j, lo, hi} (f64);
 *   device memory owned by the library (sg_device_free).  *applicable == 0: the form cannot take this input (not
 *   cosine-like, top_n > 64, threshold too low, pair list full): nothing is returned and the caller uses
 *   sg_spgemm_topn on its rows.  All ranks must take the same branch.  (Rows the pruned kernel cannot take -- more
 *   than 128 non-zeros -- are scored by the exact kernel inside the same pass; they do not switch the form off.)
 * sg_selfjoin_merge: the pairs of ALL ranks, concatenated in any order, merged into the rows of the range of `res`:
 *   afterwards these rows equal the rows sg_spgemm_topn(A, Bt) would give, bit for bit.
 * When the index is built over the row permutation (the default, sg_postings_build_flags) a range is a range of
 *   POSITIONS: the rows it covers are orig_of[row_lo .. row_hi) (sg_postings_permutation; null tables = row order).
 *   Pass the same Bt to sg_selfjoin_merge.
 * row_step (round 3): a rank's share need not be contiguous -- with row_step = s it is the rows (positions) row_hi - 1,
 *   row_hi - 1 - s, row_hi - 1 - 2 s, ... >= row_lo.  Rank r of N takes (0, n - r, N): every rank then holds rows of every
 *   cost, the shares are equal without a cost model, and a share ends with its CHEAPEST rows like the whole pass does
 *   (a contiguous range of high positions ends with rows as expensive as its first: + 0.8 ms per range at 663 k).
 *   row_step <= 1: the contiguous range [row_lo, row_hi).  Pass the same triple to sg_selfjoin_merge / sg_topn_expand_range. */
int sg_selfjoin_range(sg_ctx *ctx, const sg_csr *A, const sg_postings *Bt, int32_t top_n, double threshold,
                      int64_t row_lo, int64_t row_hi, sg_topn **out, int32_t **d_pairs, int64_t *n_pairs,
                      int32_t *pair_words, int32_t *applicable, int64_t row_step);
int sg_selfjoin_merge(sg_ctx *ctx, sg_topn *res, const sg_postings *Bt, const int32_t *d_pairs, int64_t n_pairs,
                      int32_t pair_words, int64_t row_lo, int64_t row_hi, int64_t row_step);
/* Identical rows.  sg_postings_build indexes ONE representative per group of identical right-hand rows (identical
 * strings; option SG_COLLAPSE=0 switches it off) and sg_spgemm_topn expands its result to the caller's columns, so a
 * single GPU never sees the groups.  The self-join form over ranges does: with such an index
 *   - the ranges of sg_selfjoin_range are ranges of the GROUPS' positions, [0, *n_index_rows) of sg_postings_rows; the
 *     result objects and the permutation tables have one row per group (groups are numbered by ascending lowest member;
 *     *d_group_of_row: the group of every caller row, device memory of the index, null when nothing was grouped);
 *   - after sg_selfjoin_merge, sg_topn_expand_groups turns the rows of a rank's groups into the result rows of their
 *     MEMBERS: output row k is the caller's row d_rows[k] (device array of n_rows row numbers, each a member of a group
 *     whose row of `groups` is final; null = all rows), columns are the caller's -- bit for bit the rows
 *     sg_spgemm_topn(A, Bt) returns (sorted by score descending, then column ascending).  The tables it needs (groups'
 *     members) are part of the index every rank builds, so a rank expands its own groups without any exchange.
 * (The reference has no analogue: sparse_dot_topn multiplies every duplicate row again, string_grouper.py:728-752.) */
int sg_postings_rows(const sg_postings *Bt, int64_t *n_index_rows, int64_t *n_caller_rows, const uint32_t **d_group_of_row);
/* Bytes of the index the PRUNED multiply reads while it runs -- filter postings + their segment-end tables, the 8-bit
 * copies of the rows (second filter: the header and the 16-byte units a row's entries reach -- what is written and read, not
 * the 256 bytes a record is allocated at) and the packed rows (exact scoring) -- so that a measurement can say whether they
 * fit the 256 MiB Infinity Cache (bench.py: roofline.l3_resident).  0 for an index the pruned multiply does not use. */
int sg_postings_bytes(const sg_postings *Bt, int64_t *pruned_multiply_bytes);
int sg_topn_expand_groups(sg_ctx *ctx, const sg_postings *Bt, const sg_topn *groups, const int32_t *d_rows, int64_t n_rows,
                          sg_topn **out);
/* ... the same for the rows of the groups at the positions [pos_lo, pos_hi) -- a rank's range: the library makes the list
 * (ascending row numbers; *d_rows, *n_rows: device memory of the library, sg_device_free) and expands it. */
int sg_topn_expand_range(sg_ctx *ctx, const sg_postings *Bt, const sg_topn *groups, int64_t pos_lo, int64_t pos_hi,
                         sg_topn **out, int32_t **d_rows, int64_t *n_rows, int64_t pos_step);
/* device tables of Bt's row permutation, one entry per index row: position -> row, row -> position (both null: none) */
int sg_postings_permutation(const sg_postings *Bt, const uint32_t **d_orig_of, const uint32_t **d_pos_of);
int sg_device_free(sg_ctx *ctx, void *d_ptr);

/* ------------------------------------------------------------------ measurement */
enum { SG_K_TOKENIZE = 0, SG_K_WEIGHT = 1, SG_K_POSTINGS = 2, SG_K_SPGEMM = 3 /* the multiply's whole launch group */,
       SG_K_ZIP = 4, SG_K_VOCAB = 5, SG_K_SPGEMM_KERNEL = 6 /* its dominant kernel alone (pruned: the pruned kernel without
       the pair-list pass; exact: the exact launches) */, SG_K_COUNT = 7 };
typedef struct {
    float ms[SG_K_COUNT];     /* HIP-event time of the most recent launch group of each kernel      */
    int64_t macs;             /* intermediate products of the most recent sg_spgemm_topn            */
    int64_t spgemm_bytes;     /* its algorithmic bytes (stream model)                    */
    int64_t out_nnz;          /* entries kept by the most recent sg_spgemm_topn                     */
    /* the most recent multiply, when it took the pruned kernel (all zero otherwise):                */
    int64_t prune_rows;       /* left rows it processed                                             */
    int64_t prune_postings;   /* postings it streamed (of `macs` the exact kernel would)            */
    int64_t prune_survivors;  /* candidate pairs its first filter recorded (postings' upper bounds) */
    int64_t exact_rows;       /* left rows it handed to the exact kernel                            */
    int64_t prune_bytes;      /* its algorithmic bytes: 4 per posting streamed + the 8-bit copy of  */
                              /* a row per survivor + one packed row of B per pair scored exactly   */
                              /* (without the second filter: packed row + its two pointers per      */
                              /* survivor) + A + out                                    */
    int64_t prune_symmetric;  /* != 0: self-join form (every pair scored once, from the row with    */
                              /* the larger index, then both rows' lists built from the pair list)  */
    int64_t prune_scored;     /* pairs it scored EXACTLY: the survivors the second filter (an 8-bit */
                              /* copy of the candidate's row, round 5) let through; without that    */
                              /* filter (SG_Q8=0, terms beyond 24 bits) = prune_survivors           */
} sg_stats;
/* Waits for the recorded events, so it is a synchronisation point. */
int sg_ctx_stats(sg_ctx *ctx, sg_stats *out);

#ifdef __cplusplus
}
#endif
#endif /* SG_HIP_H */
