// Host-side helpers of the public API's frames (libsg_host.so, plain C + OpenMP, loaded with ctypes.PyDLL: the caller keeps
// the GIL, so no other Python thread runs while these touch Python objects).
//
// match_strings() at 663 k names spent 44 of its 91 ms in two numpy `take`s over an object array (the left and right strings
// of 2.1 M match rows: a random access to a PyObject header per element to raise its reference count, one thread) and 18 ms
// in the conversion of the string column to UTF-8 bytes + offsets (profiles/r06_e2e_profile.log).  Both are gathers over
// immutable objects; neither needs the interpreter:
//   * sg_host_gather_objects: dst[i] = src[idx[i]] by T threads, then the reference counts raised by T threads, every
//     thread owning the OBJECTS whose address hashes to it, so that no two threads ever touch the same counter;
//   * sg_host_ascii_lengths / sg_host_ascii_copy: lengths, then bytes, of a column of compact-ASCII str objects (read-only
//     on the objects).  Anything else in the column -- a non-ASCII str, a non-str -- is reported and the caller takes the
//     general path (pyarrow), which also raises the reference's TypeError for non-strings.
// Replaces nothing of the reference's arithmetic: string_grouper/string_grouper.py:987-995 (the frames), :351-362 (the type
// check) are pandas code in the reference too.
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>
#include <string.h>
#include <omp.h>

// 0 ok; 1 an index is out of range; 2 dst does not hold n references to None (a fresh np.empty(n, object) does)
// Two passes: the pointers (every thread a run of positions: a plain gather), then the reference counts -- every thread
// walks ALL of dst and counts the objects whose address hashes to it, so that one object is only ever counted by one
// thread however many positions (of src and of dst) hold it.
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
#pragma omp parallel num_threads(threads)
    {
#pragma omp for schedule(static)
        for (int64_t i = 0; i < n; ++i) dst[i] = src[idx[i]];
        const uint64_t t = (uint64_t)omp_get_thread_num(), T = (uint64_t)omp_get_num_threads();
        for (int64_t i = 0; i < n; ++i) {
            PyObject *o = dst[i];
            if ((((uint64_t)(uintptr_t)o >> 4) * 0x9E3779B97F4A7C15ull >> 40) % T == t) ++o->ob_refcnt;
        }
    }
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
