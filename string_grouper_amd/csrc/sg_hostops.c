// Host-side helpers of the public API's frames (libsg_host.so, plain C + OpenMP, loaded with ctypes.PyDLL: the caller keeps
// the GIL, so no other Python thread runs while these touch Python objects).
//
// match_strings() at 663 k names spent 44 of its 91 ms in two numpy `take`s over an object array (the left and right strings
// of 2.1 M match rows: a random access to a PyObject header per element to raise its reference count, one thread) and 18 ms
// in the conversion of the string column to UTF-8 bytes + offsets (profiles/r06_e2e_profile.log).  Both are gathers over
// immutable objects; neither needs the interpreter:
//   * sg_host_gather_objects: dst[i] = src[idx[i]] by T threads with a count per source position, then the reference counts
//     raised once per source position (atomic adds: one object may sit at several positions);
//   * sg_host_ascii_lengths / sg_host_ascii_copy: lengths, then bytes, of a column of compact-ASCII str objects (read-only
//     on the objects).  Anything else in the column -- a non-ASCII str, a non-str -- is reported and the caller takes the
//     general path (pyarrow), which also raises the reference's TypeError for non-strings.
// Replaces nothing of the reference's arithmetic: string_grouper/string_grouper.py:987-995 (the frames), :351-362 (the type
// check) are pandas code in the reference too.
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>

// 0 ok; 1 an index is out of range; 2 dst does not hold n references to None (a fresh np.empty(n, object) does); 3 no memory
// Three passes, none of which walks anything twice: (1) the indices are checked; (2) every thread gathers the pointers of a
// run of positions and counts, per SOURCE position, how often it was taken (atomic adds on a table of 4 bytes per source
// element: it fits the L2, and a sorted index list -- the left side of a match list -- hits the same word in runs);
// (3) every thread raises the reference counts of a run of source positions by their counts, with ATOMIC adds: one str
// object may sit at several source positions (a list that repeats a name), i.e. in two threads' runs.  The caller holds the
// GIL (ctypes.PyDLL), so nothing but these threads touches the counters meanwhile.
// (Round 6's first form raised the counts position by position, every thread walking ALL positions for the objects whose
//  address hashed to it: 16 x 2.1 M hash computations, and on a SORTED index list more threads made it slower -- 13.6 ms on
//  one thread, 101 on eight, on the build container: scripts/take_objects_bench.py has both on the GPU box's host: 8.5 - 14.7 ms -> 1.3 - 1.7 ms on sixteen threads.)
int sg_host_gather_objects(PyObject **src, int64_t n_src, const int64_t *idx, int64_t n, PyObject **dst, int threads) {
    if (n <= 0) return 0;
    if (dst[0] != Py_None || dst[n - 1] != Py_None) return 2;
    if (threads < 1) threads = 1;
    if (threads > 64) threads = 64;
    int bad = 0;
#pragma omp parallel for num_threads(threads) schedule(static) reduction(| : bad)
    for (int64_t i = 0; i < n; ++i) {
        const int64_t j = idx[i];
        if (j < 0 || j >= n_src) bad = 1;
    }
    if (bad) return 1;                   // (nothing written yet: the caller lets numpy raise)
    uint32_t *cnt = (uint32_t *)calloc((size_t)n_src + 1, sizeof(uint32_t));
    if (!cnt) return 3;
#pragma omp parallel num_threads(threads)
    {
#pragma omp for schedule(static)
        for (int64_t i = 0; i < n; ++i) {
            const int64_t j = idx[i];
            dst[i] = src[j];
            __atomic_fetch_add(&cnt[j], 1u, __ATOMIC_RELAXED);
        }
#pragma omp for schedule(static)
        for (int64_t j = 0; j < n_src; ++j)
            if (cnt[j]) __atomic_fetch_add(&src[j]->ob_refcnt, (Py_ssize_t)cnt[j], __ATOMIC_RELAXED);
    }
    free(cnt);
    Py_None->ob_refcnt -= n;            // the n references to None the fresh array held are gone
    return 0;
}

// 1 when every element is exactly a str (the reference's isinstance(x, str) test, string_grouper.py:351-362, on the common
// case; a subclass of str, or anything else, and the caller asks pandas)
int sg_host_all_exact_str(PyObject **objs, int64_t n, int threads) {
    if (threads < 1) threads = 1;
    if (threads > 64) threads = 64;
    int other = 0;
#pragma omp parallel for num_threads(threads) schedule(static) reduction(| : other)
    for (int64_t i = 0; i < n; ++i)
        if (!PyUnicode_CheckExact(objs[i])) other = 1;
    return other ? 0 : 1;
}

// offsets[0 .. n]: running byte lengths; returns 0 when every element is a compact ASCII str, 1 otherwise
int sg_host_ascii_lengths(PyObject **objs, int64_t n, int64_t *offsets, int threads) {
    if (threads < 1) threads = 1;
    if (threads > 64) threads = 64;
    int other = 0;
    int64_t part_sum[65];
    memset(part_sum, 0, sizeof(part_sum));
#pragma omp parallel num_threads(threads) reduction(| : other)
    {
        const int t = omp_get_thread_num(), T = omp_get_num_threads();
        const int64_t lo = n * t / T, hi = n * (t + 1) / T;
        int64_t s = 0;
        for (int64_t i = lo; i < hi; ++i) {
            PyObject *o = objs[i];
            if (!PyUnicode_CheckExact(o) || !PyUnicode_IS_COMPACT_ASCII(o)) {
                other = 1;
                offsets[i + 1] = 0;
                continue;
            }
            const int64_t len = (int64_t)PyUnicode_GET_LENGTH(o);
            offsets[i + 1] = len;
            s += len;
        }
        part_sum[t + 1] = s;
#pragma omp barrier
#pragma omp single
        {
            for (int q = 0; q < T; ++q) part_sum[q + 1] += part_sum[q];
        }
        int64_t run = part_sum[t];
        for (int64_t i = lo; i < hi; ++i) {
            run += offsets[i + 1];
            offsets[i + 1] = run;
        }
    }
    offsets[0] = 0;
    return other;
}

void sg_host_ascii_copy(PyObject **objs, int64_t n, const int64_t *offsets, uint8_t *out, int threads) {
    if (threads < 1) threads = 1;
    if (threads > 64) threads = 64;
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int64_t i = 0; i < n; ++i) memcpy(out + offsets[i], PyUnicode_DATA(objs[i]), (size_t)(offsets[i + 1] - offsets[i]));
}

// ---- the match list's columns as the frames want them (round 6): rows expanded from the row pointers, columns and scores
// widened to the reference's int64 / float64 -- numpy's repeat / astype on one thread were 8 of match_strings()' 33 ms at
// 2.1 M match rows.  Every thread writes a run of ROWS' worth of output (first touch by the thread that fills it).
void sg_host_expand_rows(const int64_t *row_ptr, int64_t n_rows, int64_t *out_rows, int threads) {
    if (threads < 1) threads = 1;
    if (threads > 64) threads = 64;
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int64_t r = 0; r < n_rows; ++r)
        for (int64_t p = row_ptr[r]; p < row_ptr[r + 1]; ++p) out_rows[p] = r;
}

void sg_host_widen_i32(const int32_t *src, int64_t n, int64_t *dst, int threads) {
    if (threads < 1) threads = 1;
    if (threads > 64) threads = 64;
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int64_t i = 0; i < n; ++i) dst[i] = (int64_t)src[i];
}

void sg_host_widen_f32(const float *src, int64_t n, double *dst, int threads) {
    if (threads < 1) threads = 1;
    if (threads > 64) threads = 64;
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int64_t i = 0; i < n; ++i) dst[i] = (double)src[i];
}

// dst[i] = start + src[i] * step (the picked rows' labels of a RangeIndex; start 0 / step 1: a copy)
void sg_host_affine_i64(const int64_t *src, int64_t n, int64_t start, int64_t step, int64_t *dst, int threads) {
    if (threads < 1) threads = 1;
    if (threads > 64) threads = 64;
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int64_t i = 0; i < n; ++i) dst[i] = start + src[i] * step;
}
