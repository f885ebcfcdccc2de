/* CPU restatement ("port") of sparse_dot_topn's sp_matmul_topn -- TEST INFRASTRUCTURE ONLY.
 *
 * The reference calls the third-party package sparse_dot_topn (>= 1.1.0, pyproject.toml:29;
 * source NOT under /root/reference, wheel not installed) at
 *   string_grouper/string_grouper.py:725-732 and :737-743   (sp_matmul_topn)
 *   string_grouper/string_grouper.py:746                    (zip_sp_matmul_topn).
 * This file restates its published algorithm: row-wise Gustavson product with a dense
 * per-thread accumulator plus a linked list of touched columns, values strictly greater
 * than the threshold kept, at most top_n per row, rows parallelised with OpenMP.
 *
 * PARITY UNPINNED for: tie-break at the top-n cut, strictness at the threshold, within-row
 * order (see oracle/oracle.py header).  This port DEFINES strict '>' and the canonical
 * order (score descending, column ascending), exactly as oracle.py does; tests assert
 * port == oracle.py bit-for-bit.
 *
 * tie_rule = 1 is a SECOND, deliberately different plausible behaviour of the absent wheel, kept
 * to show what the unpinned tie-break can and cannot change (tests/test_tie_rules.py): the touched
 * columns are visited in the order of the linked list (last touched first) and a full result
 * list only admits a candidate that is STRICTLY greater than its current minimum, equal values
 * keeping their order of arrival -- which of several equal scores survives the cut then depends
 * on the order in which the columns were touched, not on the column index.  The two rules agree
 * on every entry whose score differs from the row's cut score (oracle.compare_tie_aware).
 *
 * Arithmetic: each C[i][j] is accumulated over k in ascending stored order of row i of A,
 * product and sum rounded separately in the value type (built with -ffp-contract=off), which
 * is bit-identical to scipy's csr_matmat.
 *
 * Used by: tests (fast oracle at sizes where scipy's A@B is too slow / too big) and by
 * bench.py's cpu_baseline leg ("kind": "port").  Never by the product path.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define SDTN_DEFINE(NAME, T)                                                                      \
    /* returns 0 on success, 1 on allocation failure */                                            \
    int NAME(int64_t n_left, int64_t n_right, const int64_t *a_indptr, const int32_t *a_indices,    \
             const T *a_data, const int64_t *bt_indptr, const int32_t *bt_indices,                 \
             const T *bt_data, int32_t top_n, T threshold, int32_t sort, int32_t n_threads,        \
             int32_t *out_cols, T *out_vals, int32_t *out_cnt, int32_t tie_rule)                   \
    {                                                                                              \
        int failed = 0;                                                                            \
        if (n_threads < 1) n_threads = 1;                                                          \
        _Pragma("omp parallel num_threads(n_threads)")                                             \
        {                                                                                          \
            T *sums = (T *)calloc((size_t)(n_right > 0 ? n_right : 1), sizeof(T));                 \
            int32_t *next = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n_right > 0 ? n_right : 1)); \
            if (!sums || !next) {                                                                  \
                _Pragma("omp atomic write") failed = 1;                                            \
            } else {                                                                               \
                for (int64_t j = 0; j < n_right; ++j) next[j] = -1;                                \
            }                                                                                      \
            _Pragma("omp barrier")                                                                 \
            if (!failed) {                                                                         \
                _Pragma("omp for schedule(dynamic, 64)")                                           \
                for (int64_t i = 0; i < n_left; ++i) {                                             \
                    int32_t head = -2;                                                             \
                    for (int64_t p = a_indptr[i]; p < a_indptr[i + 1]; ++p) {                      \
                        const int32_t k = a_indices[p];                                            \
                        const T a = a_data[p];                                                     \
                        for (int64_t q = bt_indptr[k]; q < bt_indptr[k + 1]; ++q) {                \
                            const int32_t j = bt_indices[q];                                       \
                            const T prod = a * bt_data[q];                                         \
                            sums[j] = sums[j] + prod;                                              \
                            if (next[j] == -1) { next[j] = head; head = j; }                       \
                        }                                                                          \
                    }                                                                              \
                    int32_t *oc = out_cols + (size_t)i * (size_t)top_n;                            \
                    T *ov = out_vals + (size_t)i * (size_t)top_n;                                  \
                    int32_t cnt = 0;                                                               \
                    while (head != -2) {                                                           \
                        const int32_t j = head;                                                    \
                        const T v = sums[j];                                                       \
                        head = next[j];                                                            \
                        next[j] = -1;                                                              \
                        sums[j] = (T)0;                                                            \
                        if (!(v > threshold)) continue;                                            \
                        /* bounded insertion, order: value desc then column asc */                 \
                        int32_t pos = cnt;                                                         \
                        if (tie_rule == 1) { while (pos > 0 && ov[pos - 1] < v) --pos; }           \
                        else                                                                       \
                        while (pos > 0 && (ov[pos - 1] < v || (ov[pos - 1] == v && oc[pos - 1] > j))) --pos; \
                        if (pos >= top_n) continue;                                                \
                        int32_t last = cnt < top_n ? cnt : top_n - 1;                              \
                        for (int32_t m = last; m > pos; --m) { ov[m] = ov[m - 1]; oc[m] = oc[m - 1]; } \
                        ov[pos] = v; oc[pos] = j;                                                  \
                        if (cnt < top_n) ++cnt;                                                    \
                    }                                                                              \
                    if (!sort && cnt > 1) { /* ascending column order */                           \
                        for (int32_t x = 1; x < cnt; ++x) {                                        \
                            int32_t cj = oc[x]; T cv = ov[x]; int32_t y = x - 1;                   \
                            while (y >= 0 && oc[y] > cj) { oc[y + 1] = oc[y]; ov[y + 1] = ov[y]; --y; } \
                            oc[y + 1] = cj; ov[y + 1] = cv;                                        \
                        }                                                                          \
                    }                                                                              \
                    out_cnt[i] = cnt;                                                              \
                }                                                                                  \
            }                                                                                      \
            free(sums);                                                                            \
            free(next);                                                                            \
        }                                                                                          \
        return failed;                                                                             \
    }

SDTN_DEFINE(sdtn_sp_matmul_topn_f32, float)
SDTN_DEFINE(sdtn_sp_matmul_topn_f64, double)

/* Number of intermediate products ("MACs") of A x Bt: sum_i sum_{k in A_i} nnz(Bt_k). */
int64_t sdtn_count_macs(int64_t n_left, const int64_t *a_indptr, const int32_t *a_indices,
                        const int64_t *bt_indptr)
{
    int64_t total = 0;
    for (int64_t p = 0; p < a_indptr[n_left]; ++p) {
        const int32_t k = a_indices[p];
        total += bt_indptr[k + 1] - bt_indptr[k];
    }
    return total;
}
