// K4 -- thresholded sparse top-n multiply  C = topn_rowwise(A . B^T restricted to > threshold)
// K5 -- top-n merge of column-block results (zip)
//
// Replaces sparse_dot_topn.sp_matmul_topn / zip_sp_matmul_topn as called from
// string_grouper/string_grouper.py:725-732, :737-743 and :746 of the reference.
//
// Algorithm (row-wise Gustavson product, MI355X shape).  One 64-lane wave owns one left row i at a
// time and a private accumulator tile of TILE = 2^TILE_LOG2 values in LDS (single-wave workgroups:
// no barriers anywhere, LDS is allocated in wave-sized grains, and up to 160 KiB / (TILE * s) waves
// are resident per CU).  For every column tile t of the right-hand side and every non-zero (k, a) of
// row i IN ASCENDING k, the wave streams segment (k, t) of the inverted index (sg_postings.hip) with
// coalesced loads -- lane l takes posting slo + l -- and issues one LDS atomic add per lane:
//         acc[j - t*TILE] += a * b            (product rounded, then sum rounded: no FMA)
// All j inside one segment are distinct, so one wave instruction never carries two updates of the
// same accumulator, and DS instructions of one wave execute in issue order, so every accumulator
// receives its products in ascending k: the same order scipy / sparse_dot_topn use, hence bit-equal
// scores (a requirement for bit-equal match indices next to the threshold and at the top-n cut).
// After the last k the wave sweeps its tile with 16-byte LDS reads (re-zeroing as it goes), finds
// values > threshold by ballot and inserts them into a sorted top-n list held one entry per lane
// in registers (order: score descending, then column ascending).  Rows are handed out by a global
// atomic counter (persistent waves) which balances the skew of name data (a few n-grams are present
// in 20 % of all rows).
//
// Column tiles are processed in groups by separate launches ("tile groups").  All waves of one
// launch stream the same few MB of postings, so these stay resident in each XCD's 4 MiB L2; the
// per-row top-n state lives in the output arrays between launches.
//
// Roofline: HBM/cache bandwidth bound, no MFMA (0.25 flop per byte; scatter, not a dense contraction).
// Algorithmic bytes per intermediate product ("MAC"): 4 + s (one (j, value) posting), plus
// nnz(A)*(4+s) + out -- see DESIGN.md.
#include <math.h>

#include "sg_internal.h"

#define SG_TOPN_LANES 64   // entries of the register-resident list = lanes of a wave

template <typename T>
__device__ __forceinline__ T wave_read(T v, int src_lane);   // value of v in lane src_lane (uniform src)

template <>
__device__ __forceinline__ float wave_read<float>(float v, int src_lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src_lane));
}
template <>
__device__ __forceinline__ int wave_read<int>(int v, int src_lane) {
    return __builtin_amdgcn_readlane(v, src_lane);
}
template <>
__device__ __forceinline__ uint32_t wave_read<uint32_t>(uint32_t v, int src_lane) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, src_lane);
}
template <>
__device__ __forceinline__ double wave_read<double>(double v, int src_lane) {
    const uint64_t u = __builtin_bit_cast(uint64_t, v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)u, src_lane);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), src_lane);
    return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}

template <typename T>
__device__ __forceinline__ T mul_rn(T a, T b);
template <>
__device__ __forceinline__ float mul_rn<float>(float a, float b) { return __fmul_rn(a, b); }
template <>
__device__ __forceinline__ double mul_rn<double>(double a, double b) { return __dmul_rn(a, b); }

template <typename T>
__device__ __forceinline__ void lds_add(T *p, T v) {
    // LDS float atomic add without return (ds_add_f32 / ds_add_f64): one DS instruction per 64 MACs
    (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <typename T>
struct TopList {   // lane r holds the r-th best (score, col); empty slots are (-inf, INT_MAX)
    T s;
    int c;
    __device__ __forceinline__ void clear() {
        s = -INFINITY;
        c = INT32_MAX;
    }
    // (ns, nc) are wave-uniform.  floor_*: only entries strictly after the floor key are eligible
    // (used by the passes that collect ranks 64.. of a row; floor_s = +inf disables it).
    __device__ __forceinline__ void insert(T ns, int nc, int lane) {
        const bool mine_first = (s > ns) || (s == ns && c < nc);
        const int pos = __popcll(__ballot(mine_first));
        if (pos >= SG_TOPN_LANES) return;
        const T us = __shfl_up(s, 1, 64);
        const int uc = __shfl_up(c, 1, 64);
        if (lane > pos) {
            s = us;
            c = uc;
        } else if (lane == pos) {
            s = ns;
            c = nc;
        }
    }
};

// posting entry: f32 -> packed {int32 row, float value} (one 8-byte load per lane);
//                f64 -> rows[] + vals[] (4 + 8 bytes)
template <typename T>
struct Post;
template <>
struct Post<float> {
    typedef uint2 reg_t;
    static __device__ __forceinline__ reg_t load(const int32_t *, const float *vals, uint32_t idx) {
        return reinterpret_cast<const uint2 *>(vals)[idx];
    }
    static __device__ __forceinline__ int row(const reg_t &r) { return (int)r.x; }
    static __device__ __forceinline__ float val(const reg_t &r) { return __uint_as_float(r.y); }
};
template <>
struct Post<double> {
    struct reg_t {
        int j;
        double v;
    };
    static __device__ __forceinline__ reg_t load(const int32_t *rows, const double *vals, uint32_t idx) {
        reg_t r;
        r.j = rows[idx];
        r.v = vals[idx];
        return r;
    }
    static __device__ __forceinline__ int row(const reg_t &r) { return r.j; }
    static __device__ __forceinline__ double val(const reg_t &r) { return r.v; }
};

// The postings a row needs from one column tile -- up to 64 segments (one per non-zero of the row,
// lane l owns segment l: [lo_l, lo_l + len_l)) -- are walked as ONE flat list of S = sum(len) entries
// in ascending segment (= ascending k) order, 64 entries ("a window") per step, every lane busy.
// Windows are processed in batches: the loads of a whole batch (DEPTH windows = DEPTH x 512 B per
// wave) are issued back to back in straight-line code and only then consumed, so DEPTH loads are in
// flight per wave and the compiler emits counted vmcnt waits (15, 14, ...) while consuming.  This is
// what hides the ~1-2 us L2/MALL latency: the first version of this kernel waited for every posting
// segment separately and ran at 1.5 TB/s algorithmic.
// Consuming a window keeps the bit-exact order: lanes of one segment carry distinct columns, so each
// segment present in the window gets its own exec-masked ds_add, issued in ascending k.
template <typename T, int TILE>
struct FlatWalk {
    T *acc;
    const int32_t *__restrict__ post_rows;
    const T *__restrict__ post_vals;
    uint32_t end;    // inclusive prefix sum of the segment lengths (lane l: end of segment l)
    uint32_t base;   // lo_l - start_l: posting index of flat position p in segment l is base_l + p
    T a;             // lane l: value of the row's non-zero l
    uint32_t S;      // total entries
    uint32_t sb;     // first segment whose end lies beyond the next window (uniform)
    int nseg;
    int lane;

    __device__ __forceinline__ void locate(uint32_t w, uint32_t &s, uint32_t &idx) {
        const uint32_t p0 = w << 6;
        const uint32_t p = p0 + lane;
        const uint32_t last = p0 + 63;
        s = sb;
        uint32_t q = sb;
        while (q < (uint32_t)nseg) {   // segment ends that fall inside this window
            const uint32_t e = wave_read<uint32_t>(end, (int)q);
            if (e > last) break;
            s += (p >= e) ? 1u : 0u;
            ++q;
        }
        sb = q;
        const bool valid = p < S;
        if (!valid) s = 0;
        const uint32_t bs = __shfl(base, (int)s, 64);
        idx = bs + p;
        if (!valid) {          // idle lanes of the last window: read entry lo_0 (in bounds: the arrays are
            idx = bs;          // padded by 64 entries) and stay masked out of the adds
            s |= 0x80000000u;
        }
    }

    __device__ __forceinline__ void consume(const typename Post<T>::reg_t &r, uint32_t s) {
        const bool valid = (s & 0x80000000u) == 0;
        const uint32_t sl = s & 63u;
        const T as = __shfl(a, (int)sl, 64);
        const T prod = mul_rn<T>(as, Post<T>::val(r));
        const int j = Post<T>::row(r);
        uint64_t pend = __ballot(valid);
        while (pend) {   // one masked DS instruction per segment present, ascending k
            const int f = __builtin_ctzll(pend);
            const uint32_t sc = wave_read<uint32_t>(sl, f);
            const bool mine = valid && sl == sc;
            if (mine) lds_add<T>(&acc[j & (TILE - 1)], prod);
            pend &= ~__ballot(mine);
        }
    }

    template <int B>
    __device__ __forceinline__ void batch(uint32_t w) {
        uint32_t s[B], idx[B];
        typename Post<T>::reg_t r[B];
#pragma unroll
        for (int b = 0; b < B; ++b) locate(w + b, s[b], idx[b]);
#pragma unroll
        for (int b = 0; b < B; ++b) r[b] = Post<T>::load(post_rows, post_vals, idx[b]);
#pragma unroll
        for (int b = 0; b < B; ++b) consume(r[b], s[b]);
    }
};

template <typename T, int TILE, int DEPTH>
__device__ __forceinline__ void accumulate_flat(T *acc, const int32_t *__restrict__ post_rows,
                                                const T *__restrict__ post_vals, uint32_t lo, uint32_t len,
                                                T a, int nseg, int lane) {
    // inclusive scan of the segment lengths over the lanes
    uint32_t end = len;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(end, d, 64);
        if (lane >= d) end += up;
    }
    const uint32_t S = wave_read<uint32_t>(end, 63);
    if (S == 0) return;
    FlatWalk<T, TILE> fw;
    fw.acc = acc;
    fw.post_rows = post_rows;
    fw.post_vals = post_vals;
    fw.end = end;
    fw.base = lo - (end - len);
    fw.a = a;
    fw.S = S;
    fw.sb = 0;
    fw.nseg = nseg;
    fw.lane = lane;
    const uint32_t W = (S + 63) >> 6;   // windows of this (row, tile)
    uint32_t w = 0;
    while (W - w >= (uint32_t)DEPTH) {
        fw.template batch<DEPTH>(w);
        w += DEPTH;
    }
    if (DEPTH > 4)
        while (W - w >= 4u) {
            fw.template batch<4>(w);
            w += 4;
        }
    while (w < W) {
        fw.template batch<1>(w);
        w += 1;
    }
}

template <typename T, int TILE_LOG2, int DEPTH>
__global__ void __launch_bounds__(64)
spgemm_topn_kernel(const int64_t *__restrict__ a_indptr, const int32_t *__restrict__ a_indices,
                   const T *__restrict__ a_data, uint32_t n_left, const uint32_t *__restrict__ seg,
                   const int32_t *__restrict__ post_rows, const T *__restrict__ post_vals, int32_t n_tiles,
                   int32_t tile_begin, int32_t tile_end, int32_t keep /* <= 64 entries this pass */,
                   int32_t pass_off /* 64 * pass */, int32_t out_stride, T thr, int32_t *__restrict__ out_cols,
                   T *__restrict__ out_vals, int32_t *__restrict__ out_cnt, uint32_t *row_counter) {
    constexpr int TILE = 1 << TILE_LOG2;
    constexpr int VEC = 16 / sizeof(T);   // values per 16-byte LDS access
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T *acc = reinterpret_cast<T *>(smem);
    const int lane = threadIdx.x;

    typedef T vec_t __attribute__((ext_vector_type(VEC)));
    vec_t *acc_v = reinterpret_cast<vec_t *>(acc);
    for (int x = lane; x < TILE / VEC; x += 64) acc_v[x] = (vec_t)(T)0;

    for (;;) {
        uint32_t row = 0;
        if (lane == 0) row = atomicAdd(row_counter, 1u);
        row = (uint32_t)__builtin_amdgcn_readfirstlane((int)row);
        if (row >= n_left) break;

        const int64_t rlo = a_indptr[row];
        const int nnz = (int)(a_indptr[row + 1] - rlo);
        const size_t obase = (size_t)row * (size_t)out_stride + (size_t)pass_off;

        // ---- restore the row's running state (written by the previous tile group / pass)
        TopList<T> top;
        top.clear();
        T floor_s = INFINITY;
        int floor_c = -1;
        int prev_cnt = 0;
        if (tile_begin > 0 || pass_off > 0) prev_cnt = out_cnt[row];
        if (pass_off > 0) {
            if (prev_cnt < pass_off) continue;              // earlier passes did not fill up: row is complete
            floor_s = out_vals[obase - 1];
            floor_c = out_cols[obase - 1];
        }
        if (tile_begin > 0) {
            const int have = prev_cnt - pass_off;           // entries collected so far in this pass
            if (lane < have) {
                top.s = out_vals[obase + lane];
                top.c = out_cols[obase + lane];
            }
        }

        if (nnz > 0) {
            // first 64 non-zeros of the row stay in registers for all tiles (lane l holds non-zero l)
            int k0 = -1;
            T a0 = (T)0;
            if (lane < nnz) {
                k0 = a_indices[rlo + lane];
                a0 = a_data[rlo + lane];
            }
            // segment bounds of (k0, t): lo is the previous tile's hi; the next tile's hi is fetched
            // one tile ahead so that its latency hides behind this tile's work
            uint32_t lo0 = 0, hi0 = 0, hi_next = 0;
            if (k0 >= 0) {
                const uint32_t *sp = seg + (int64_t)k0 * n_tiles + tile_begin;
                lo0 = sp[0];
                hi0 = sp[1];
            }

            for (int t = tile_begin; t < tile_end; ++t) {
                if (k0 >= 0 && t + 1 < tile_end) hi_next = seg[(int64_t)k0 * n_tiles + t + 2];
                bool touched = false;
                for (int c0 = 0; c0 < nnz; c0 += 64) {
                    T a;
                    uint32_t lo = 0, hi = 0;
                    if (c0 == 0) {
                        a = a0;
                        lo = lo0;
                        hi = hi0;
                    } else {
                        a = (T)0;
                        if (c0 + lane < nnz) {
                            const int k = a_indices[rlo + c0 + lane];
                            a = a_data[rlo + c0 + lane];
                            lo = seg[(int64_t)k * n_tiles + t];
                            hi = seg[(int64_t)k * n_tiles + t + 1];
                        }
                    }
                    if (__ballot(hi > lo) == 0) continue;
                    touched = true;
                    const int nseg = nnz - c0 < 64 ? nnz - c0 : 64;
                    accumulate_flat<T, TILE, DEPTH>(acc, post_rows, post_vals, lo, hi - lo, a, nseg, lane);
                }
                lo0 = hi0;
                hi0 = hi_next;
                if (!touched) continue;   // accumulators are still all zero

                // ---- sweep the tile: find values > thr, re-zero
                const int col_base = t << TILE_LOG2;
                for (int x0 = 0; x0 < TILE / VEC; x0 += 64) {
                    const vec_t v = acc_v[x0 + lane];
                    acc_v[x0 + lane] = (vec_t)(T)0;
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        uint64_t hm = __ballot(v[e] > thr);
                        while (hm) {
                            const int src = __builtin_ctzll(hm);
                            hm &= hm - 1;
                            const T ns = wave_read<T>(v[e], src);
                            const int nc = col_base + (x0 + src) * VEC + e;
                            if (ns < floor_s || (ns == floor_s && nc > floor_c)) top.insert(ns, nc, lane);
                        }
                    }
                }
            }
        }

        // ---- store the row's state (final when tile_end == n_tiles)
        int cnt = __popcll(__ballot(top.c != INT32_MAX));
        if (cnt > keep) cnt = keep;
        if (lane < cnt) {
            out_vals[obase + lane] = top.s;
            out_cols[obase + lane] = top.c;
        }
        if (lane == 0) out_cnt[row] = pass_off + cnt;
    }
}

// Sum over the non-zeros of A of the posting-list length of their term = number of intermediate
// products; also sums the result counts.  Measurement only (bench.py roofline).
__global__ void __launch_bounds__(256) count_macs_kernel(const int64_t *__restrict__ a_indptr,
                                                         const int32_t *__restrict__ a_indices, int64_t n_left,
                                                         const uint32_t *__restrict__ seg, int32_t n_tiles,
                                                         unsigned long long *out_macs) {
    const int64_t p0 = a_indptr[0], p1 = a_indptr[n_left];
    unsigned long long local = 0;
    for (int64_t p = p0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < p1; p += (int64_t)gridDim.x * blockDim.x) {
        const int64_t k = a_indices[p];
        local += seg[(k + 1) * n_tiles] - seg[k * n_tiles];
    }
    for (int d = 32; d > 0; d >>= 1) local += __shfl_down(local, d, 64);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(out_macs, local);
}

__global__ void __launch_bounds__(256) sum_counts_kernel(const int32_t *__restrict__ cnt, int64_t n,
                                                         unsigned long long *out) {
    unsigned long long local = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        local += (unsigned long long)cnt[i];
    for (int d = 32; d > 0; d >>= 1) local += __shfl_down(local, d, 64);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(out, local);
}

// Re-order every row of a fixed-stride result by ascending column (sort == 0).  One wave per row;
// dynamic LDS: stride * (4 + sizeof(T)) bytes.
template <typename T>
__global__ void __launch_bounds__(64) topn_sort_by_col_kernel(int32_t *cols, T *vals, const int32_t *cnt,
                                                              int64_t n_rows, int32_t stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T *sv = reinterpret_cast<T *>(smem);
    int32_t *sc = reinterpret_cast<int32_t *>(smem + sizeof(T) * (size_t)stride);
    const int lane = threadIdx.x;
    for (int64_t row = blockIdx.x; row < n_rows; row += gridDim.x) {
        const int n = cnt[row];
        int32_t *rc = cols + (size_t)row * stride;
        T *rv = vals + (size_t)row * stride;
        for (int i = lane; i < n; i += 64) {   // rank by counting; columns of one row are distinct
            const int myc = rc[i];
            int rank = 0;
            for (int q = 0; q < n; ++q) rank += (rc[q] < myc);
            sc[rank] = myc;
            sv[rank] = rv[i];
        }
        __syncthreads();
        for (int i = lane; i < n; i += 64) {
            rc[i] = sc[i];
            rv[i] = sv[i];
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// K5: merge of column-block results (zip_sp_matmul_topn).  One wave per row, register top-n.
template <typename T>
struct ZipPart {
    const int32_t *cols;
    const T *vals;
    const int32_t *cnt;
    int32_t stride;
    int32_t col_offset;
};

template <typename T>
__global__ void __launch_bounds__(64) topn_zip_kernel(const ZipPart<T> *__restrict__ parts, int32_t n_parts,
                                                      int64_t n_rows, int32_t keep, int32_t pass_off,
                                                      int32_t out_stride, int32_t *out_cols, T *out_vals,
                                                      int32_t *out_cnt) {
    const int lane = threadIdx.x;
    for (int64_t row = blockIdx.x; row < n_rows; row += gridDim.x) {
        const size_t obase = (size_t)row * out_stride + pass_off;
        TopList<T> top;
        top.clear();
        T floor_s = INFINITY;
        int floor_c = -1;
        if (pass_off > 0) {
            if (out_cnt[row] < pass_off) continue;
            floor_s = out_vals[obase - 1];
            floor_c = out_cols[obase - 1];
        }
        for (int b = 0; b < n_parts; ++b) {
            const ZipPart<T> part = parts[b];
            const int n = part.cnt[row];
            for (int base = 0; base < n; base += 64) {
                T v = (T)0;
                int c = 0;
                const bool ok = base + lane < n;
                if (ok) {
                    v = part.vals[(size_t)row * part.stride + base + lane];
                    c = part.cols[(size_t)row * part.stride + base + lane] + part.col_offset;
                }
                uint64_t m = __ballot(ok);
                while (m) {
                    const int src = __builtin_ctzll(m);
                    m &= m - 1;
                    const T ns = wave_read<T>(v, src);
                    const int nc = wave_read<int>(c, src);
                    if (ns < floor_s || (ns == floor_s && nc > floor_c)) top.insert(ns, nc, lane);
                }
            }
        }
        int cnt = __popcll(__ballot(top.c != INT32_MAX));
        if (cnt > keep) cnt = keep;
        if (lane < cnt) {
            out_vals[obase + lane] = top.s;
            out_cols[obase + lane] = top.c;
        }
        if (lane == 0) out_cnt[row] = pass_off + cnt;
    }
}

// ================================================================================================
// host side
// ================================================================================================
static int env_int(const char *name, int dflt) {
    const char *v = getenv(name);
    if (!v || !*v) return dflt;
    return atoi(v);
}

template <typename T, int TILE_LOG2, int DEPTH>
static int launch_spgemm(sg_ctx *ctx, const sg_csr *A, const sg_postings *Bt, int32_t tile_begin, int32_t tile_end,
                         int32_t keep, int32_t pass_off, sg_topn *r, T thr, uint32_t *counter, unsigned grid) {
    const size_t lds = sizeof(T) << TILE_LOG2;
    auto kern = spgemm_topn_kernel<T, TILE_LOG2, DEPTH>;
    if (lds > 48 * 1024) {
        static bool done = false;   // per instantiation
        if (!done) {
            SG_HIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            done = true;
        }
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64), lds, ctx->stream, A->d_indptr, A->d_indices, (const T *)A->d_data,
                       (uint32_t)A->n_rows, (const uint32_t *)Bt->d_seg, (const int32_t *)Bt->d_rows,
                       (const T *)Bt->d_vals, Bt->n_tiles, tile_begin, tile_end, keep, pass_off, r->stride, thr,
                       r->d_cols, (T *)r->d_vals, r->d_counts, counter);
    SG_HIP_TRY(hipGetLastError());
    return SG_OK;
}

template <typename T, int TILE_LOG2>
static int dispatch_depth(sg_ctx *ctx, const sg_csr *A, const sg_postings *Bt, int32_t tile_begin, int32_t tile_end,
                          int32_t keep, int32_t pass_off, sg_topn *r, T thr, uint32_t *counter, unsigned grid,
                          int depth) {
    switch (depth) {
        case 4: return launch_spgemm<T, TILE_LOG2, 4>(ctx, A, Bt, tile_begin, tile_end, keep, pass_off, r, thr, counter, grid);
        case 16: return launch_spgemm<T, TILE_LOG2, 16>(ctx, A, Bt, tile_begin, tile_end, keep, pass_off, r, thr, counter, grid);
        default: return launch_spgemm<T, TILE_LOG2, 8>(ctx, A, Bt, tile_begin, tile_end, keep, pass_off, r, thr, counter, grid);
    }
}

template <typename T>
static int dispatch_spgemm(sg_ctx *ctx, const sg_csr *A, const sg_postings *Bt, int32_t tile_begin, int32_t tile_end,
                           int32_t keep, int32_t pass_off, sg_topn *r, T thr, uint32_t *counter, unsigned grid) {
    const int depth = env_int("SG_DEPTH", 8);
    switch (Bt->tile_log2) {
        case 10: return dispatch_depth<T, 10>(ctx, A, Bt, tile_begin, tile_end, keep, pass_off, r, thr, counter, grid, depth);
        case 11: return dispatch_depth<T, 11>(ctx, A, Bt, tile_begin, tile_end, keep, pass_off, r, thr, counter, grid, depth);
        case 12: return dispatch_depth<T, 12>(ctx, A, Bt, tile_begin, tile_end, keep, pass_off, r, thr, counter, grid, depth);
        case 13: return dispatch_depth<T, 13>(ctx, A, Bt, tile_begin, tile_end, keep, pass_off, r, thr, counter, grid, depth);
        default:
            sg_set_error("postings tile of 2^%d columns is not supported by the multiply (2^10..2^13)", Bt->tile_log2);
            return SG_ERR_UNSUPPORTED;
    }
}

static int topn_alloc(sg_ctx *ctx, int64_t n_rows, int64_t n_cols, int32_t stride, int32_t dtype, sg_topn **out) {
    sg_topn *r = new (std::nothrow) sg_topn();
    if (!r) return SG_ERR_OOM;
    r->ctx = ctx;
    r->n_rows = n_rows;
    r->n_cols = n_cols;
    r->stride = stride;
    r->dtype = dtype;
    const size_t cells = (size_t)n_rows * (size_t)stride + 64;
    int st = sg_alloc(ctx, cells, &r->d_cols);
    if (st == SG_OK) st = ctx->alloc(cells * (dtype == SG_F64 ? 8 : 4), &r->d_vals);
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)n_rows + 64, &r->d_counts);
    if (st != SG_OK) {
        sg_topn_free(r);
        return st;
    }
    *out = r;
    return SG_OK;
}

extern "C" int sg_spgemm_topn(sg_ctx *ctx, const sg_csr *A, const sg_postings *Bt, int32_t top_n, double threshold,
                              int32_t sort, sg_topn **out) {
    SG_REQUIRE(ctx && A && Bt && out, "null argument");
    SG_REQUIRE(A->n_cols == Bt->n_terms, "A and B have different numbers of columns (vocabulary size)");
    SG_REQUIRE(A->dtype == Bt->dtype, "A and B have different value types");
    SG_REQUIRE(top_n >= 1, "top_n must be >= 1");
    // only touched accumulators are candidates; with non-negative data (TF-IDF) touched == value > 0,
    // so a negative threshold selects the same entries as 0
    if (!(threshold > 0.0)) threshold = 0.0;
    int64_t stride64 = top_n;
    if (stride64 > Bt->n_right) stride64 = Bt->n_right > 0 ? Bt->n_right : 1;
    if ((double)A->n_rows * (double)stride64 > 2.0e9) {
        sg_set_error("result of %lld rows x top_n %lld does not fit the 32-bit result index; split the left matrix",
                     (long long)A->n_rows, (long long)stride64);
        return SG_ERR_OVERFLOW;
    }
    const int32_t stride = (int32_t)stride64;
    sg_topn *r = nullptr;
    SG_TRY(topn_alloc(ctx, A->n_rows, Bt->n_right, stride, A->dtype, &r));
    const size_t s = A->dtype == SG_F64 ? 8 : 4;

    // tiles per launch: keep one launch's postings (~ nnz*(4+s)/n_tiles per tile) near 2 MiB so that
    // they stay in every XCD's 4 MiB L2 while all rows stream over them
    int group = env_int("SG_TILE_GROUP", 0);
    if (group <= 0) {
        const double per_tile = (double)Bt->nnz * (double)(4 + s) / (double)(Bt->n_tiles > 0 ? Bt->n_tiles : 1);
        group = (int)(2.0 * 1024 * 1024 / (per_tile > 1.0 ? per_tile : 1.0));
        if (group < 1) group = 1;
    }
    if (group > Bt->n_tiles) group = Bt->n_tiles;
    const int n_groups = (Bt->n_tiles + group - 1) / group;
    const int n_pass = (stride + SG_TOPN_LANES - 1) / SG_TOPN_LANES;
    const int n_launch = n_groups * n_pass;

    const size_t lds = s << Bt->tile_log2;
    int waves_per_cu = (int)(ctx->lds_per_cu / lds);
    if (waves_per_cu > 32) waves_per_cu = 32;
    if (waves_per_cu < 1) waves_per_cu = 1;
    waves_per_cu = env_int("SG_WAVES_PER_CU", waves_per_cu);
    unsigned grid = (unsigned)ctx->num_cu * (unsigned)waves_per_cu;
    if ((int64_t)grid > A->n_rows) grid = (unsigned)(A->n_rows > 0 ? A->n_rows : 1);

    uint32_t *counters = nullptr;
    int st = sg_alloc(ctx, (size_t)n_launch + 1, &counters);
    if (st != SG_OK) {
        sg_topn_free(r);
        return st;
    }
    {
        SgTimer timer(ctx, SG_K_SPGEMM);
        st = SG_OK;
        if (hipMemsetAsync(counters, 0, sizeof(uint32_t) * (size_t)(n_launch + 1), ctx->stream) != hipSuccess ||
            hipMemsetAsync(r->d_counts, 0, sizeof(int32_t) * (size_t)A->n_rows, ctx->stream) != hipSuccess)
            st = SG_ERR_HIP;
        int li = 0;
        for (int pass = 0; pass < n_pass && st == SG_OK && A->n_rows > 0; ++pass) {
            const int pass_off = pass * SG_TOPN_LANES;
            const int keep = stride - pass_off < SG_TOPN_LANES ? stride - pass_off : SG_TOPN_LANES;
            for (int g = 0; g < n_groups && st == SG_OK; ++g, ++li) {
                const int tb = g * group;
                const int te = tb + group < Bt->n_tiles ? tb + group : Bt->n_tiles;
                if (A->dtype == SG_F64)
                    st = dispatch_spgemm<double>(ctx, A, Bt, tb, te, keep, pass_off, r, (double)threshold,
                                                 counters + li, grid);
                else
                    st = dispatch_spgemm<float>(ctx, A, Bt, tb, te, keep, pass_off, r, (float)threshold,
                                                counters + li, grid);
            }
        }
    }
    if (st == SG_OK && !sort && A->n_rows > 0) {
        unsigned g2 = (unsigned)(A->n_rows < 65535 * 16 ? A->n_rows : 65535 * 16);
        const size_t l2 = (size_t)stride * (4 + s);
        if (l2 > 64 * 1024) {
            sg_set_error("sort=0 with top_n=%d is not supported", stride);
            st = SG_ERR_UNSUPPORTED;
        } else if (A->dtype == SG_F64)
            hipLaunchKernelGGL(topn_sort_by_col_kernel<double>, dim3(g2), dim3(64), l2, ctx->stream, r->d_cols,
                               (double *)r->d_vals, r->d_counts, A->n_rows, stride);
        else
            hipLaunchKernelGGL(topn_sort_by_col_kernel<float>, dim3(g2), dim3(64), l2, ctx->stream, r->d_cols,
                               (float *)r->d_vals, r->d_counts, A->n_rows, stride);
    }
    // measurement words: MACs and kept entries of this multiply
    if (st == SG_OK) {
        (void)hipMemsetAsync(ctx->d_stat_words, 0, 2 * sizeof(int64_t), ctx->stream);
        if (A->nnz > 0)
            hipLaunchKernelGGL(count_macs_kernel, dim3(1024), dim3(256), 0, ctx->stream, A->d_indptr, A->d_indices,
                               A->n_rows, (const uint32_t *)Bt->d_seg, Bt->n_tiles,
                               (unsigned long long *)ctx->d_stat_words);
        if (A->n_rows > 0)
            hipLaunchKernelGGL(sum_counts_kernel, dim3(1024), dim3(256), 0, ctx->stream, r->d_counts, A->n_rows,
                               (unsigned long long *)(ctx->d_stat_words + 1));
        // algorithmic bytes (stream model): macs*(4+s) + nnz(A)*(4+s) + (nL+V+2)*4 + out*(4+s);
        // the MAC and output terms are added in sg_ctx_stats once the device counters are read
        ctx->spgemm_entry_bytes = (int64_t)(4 + s);
        ctx->spgemm_fixed_bytes = A->nnz * (int64_t)(4 + s) + (A->n_rows + Bt->n_terms + 2) * 4;
        if (hipGetLastError() != hipSuccess) st = SG_ERR_HIP;
    }
    ctx->release(counters);
    if (st != SG_OK) {
        sg_topn_free(r);
        return st;
    }
    *out = r;
    return SG_OK;
}

extern "C" int sg_topn_dims(const sg_topn *r, int64_t *n_rows, int32_t *stride, int32_t *dtype, int64_t *n_cols) {
    SG_REQUIRE(r != nullptr, "result is null");
    if (n_rows) *n_rows = r->n_rows;
    if (stride) *stride = r->stride;
    if (dtype) *dtype = r->dtype;
    if (n_cols) *n_cols = r->n_cols;
    return SG_OK;
}

extern "C" int sg_topn_device_ptrs(const sg_topn *r, const int32_t **d_cols, const void **d_vals,
                                   const int32_t **d_counts) {
    SG_REQUIRE(r != nullptr, "result is null");
    if (d_cols) *d_cols = r->d_cols;
    if (d_vals) *d_vals = r->d_vals;
    if (d_counts) *d_counts = r->d_counts;
    return SG_OK;
}

extern "C" int sg_topn_to_host(sg_ctx *ctx, const sg_topn *r, int32_t *cols, void *vals, int32_t *counts) {
    SG_REQUIRE(ctx && r && counts, "null argument");
    const size_t cells = (size_t)r->n_rows * (size_t)r->stride;
    const size_t s = r->dtype == SG_F64 ? 8 : 4;
    if (cells > 0) {
        SG_REQUIRE(cols && vals, "null output");
        SG_HIP_TRY(hipMemcpyAsync(cols, r->d_cols, cells * 4, hipMemcpyDeviceToHost, ctx->stream));
        SG_HIP_TRY(hipMemcpyAsync(vals, r->d_vals, cells * s, hipMemcpyDeviceToHost, ctx->stream));
    }
    if (r->n_rows > 0)
        SG_HIP_TRY(hipMemcpyAsync(counts, r->d_counts, (size_t)r->n_rows * 4, hipMemcpyDeviceToHost, ctx->stream));
    SG_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SG_OK;
}

extern "C" int sg_topn_from_host(sg_ctx *ctx, int64_t n_rows, int64_t n_cols, int32_t stride, int32_t dtype,
                                 const int32_t *cols, const void *vals, const int32_t *counts, sg_topn **out) {
    SG_REQUIRE(ctx && counts && out && n_rows >= 0 && stride >= 1, "bad argument");
    SG_REQUIRE(dtype == SG_F32 || dtype == SG_F64, "dtype must be SG_F32 or SG_F64");
    sg_topn *r = nullptr;
    SG_TRY(topn_alloc(ctx, n_rows, n_cols, stride, dtype, &r));
    const size_t cells = (size_t)n_rows * (size_t)stride;
    const size_t s = dtype == SG_F64 ? 8 : 4;
    hipError_t e = hipSuccess;
    if (cells > 0) {
        e = hipMemcpyAsync(r->d_cols, cols, cells * 4, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(r->d_vals, vals, cells * s, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(r->d_counts, counts, (size_t)n_rows * 4, hipMemcpyHostToDevice, ctx->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        sg_set_error("sg_topn_from_host: %s", hipGetErrorString(e));
        sg_topn_free(r);
        return SG_ERR_HIP;
    }
    *out = r;
    return SG_OK;
}

extern "C" int sg_topn_free(sg_topn *r) {
    if (!r) return SG_OK;
    r->ctx->release(r->d_cols);
    r->ctx->release(r->d_vals);
    r->ctx->release(r->d_counts);
    delete r;
    return SG_OK;
}

extern "C" int sg_topn_zip(sg_ctx *ctx, const sg_topn *const *parts, const int64_t *col_offsets, int32_t n_parts,
                           int32_t top_n, sg_topn **out) {
    SG_REQUIRE(ctx && parts && col_offsets && out && n_parts >= 1 && top_n >= 1, "bad argument");
    const int64_t n_rows = parts[0]->n_rows;
    const int32_t dtype = parts[0]->dtype;
    int64_t total_cols = 0, total_stride = 0;
    for (int b = 0; b < n_parts; ++b) {
        SG_REQUIRE(parts[b] && parts[b]->n_rows == n_rows && parts[b]->dtype == dtype, "parts disagree in shape/dtype");
        const int64_t end = col_offsets[b] + parts[b]->n_cols;
        if (end > total_cols) total_cols = end;
        total_stride += parts[b]->stride;
    }
    if (total_cols > INT32_MAX) {
        sg_set_error("zipped column count exceeds int32");
        return SG_ERR_OVERFLOW;
    }
    int64_t stride64 = top_n < total_stride ? top_n : total_stride;
    if (stride64 < 1) stride64 = 1;
    const int32_t stride = (int32_t)stride64;
    sg_topn *r = nullptr;
    SG_TRY(topn_alloc(ctx, n_rows, total_cols, stride, dtype, &r));
    const size_t s = dtype == SG_F64 ? 8 : 4;
    // part descriptors: host-pinned scratch would add a dependency; a tiny pooled device buffer + sync copy
    std::vector<unsigned char> host_desc((size_t)n_parts * (dtype == SG_F64 ? sizeof(ZipPart<double>) : sizeof(ZipPart<float>)));
    for (int b = 0; b < n_parts; ++b) {
        if (dtype == SG_F64) {
            ZipPart<double> d{parts[b]->d_cols, (const double *)parts[b]->d_vals, parts[b]->d_counts, parts[b]->stride,
                              (int32_t)col_offsets[b]};
            memcpy(host_desc.data() + (size_t)b * sizeof(d), &d, sizeof(d));
        } else {
            ZipPart<float> d{parts[b]->d_cols, (const float *)parts[b]->d_vals, parts[b]->d_counts, parts[b]->stride,
                             (int32_t)col_offsets[b]};
            memcpy(host_desc.data() + (size_t)b * sizeof(d), &d, sizeof(d));
        }
    }
    void *d_desc = nullptr;
    int st = ctx->alloc(host_desc.size(), &d_desc);
    if (st != SG_OK) {
        sg_topn_free(r);
        return st;
    }
    (void)s;
    SG_HIP_TRY(hipMemcpyAsync(d_desc, host_desc.data(), host_desc.size(), hipMemcpyHostToDevice, ctx->stream));
    SG_HIP_TRY(hipStreamSynchronize(ctx->stream));   // host_desc is a local
    {
        SgTimer timer(ctx, SG_K_ZIP);
        SG_HIP_TRY(hipMemsetAsync(r->d_counts, 0, sizeof(int32_t) * (size_t)n_rows, ctx->stream));
        const int n_pass = (stride + SG_TOPN_LANES - 1) / SG_TOPN_LANES;
        unsigned grid = (unsigned)(n_rows < 256 * 32 ? (n_rows > 0 ? n_rows : 1) : 256 * 32);
        for (int pass = 0; pass < n_pass && n_rows > 0; ++pass) {
            const int pass_off = pass * SG_TOPN_LANES;
            const int keep = stride - pass_off < SG_TOPN_LANES ? stride - pass_off : SG_TOPN_LANES;
            if (dtype == SG_F64)
                hipLaunchKernelGGL(topn_zip_kernel<double>, dim3(grid), dim3(64), 0, ctx->stream,
                                   (const ZipPart<double> *)d_desc, n_parts, n_rows, keep, pass_off, stride, r->d_cols,
                                   (double *)r->d_vals, r->d_counts);
            else
                hipLaunchKernelGGL(topn_zip_kernel<float>, dim3(grid), dim3(64), 0, ctx->stream,
                                   (const ZipPart<float> *)d_desc, n_parts, n_rows, keep, pass_off, stride, r->d_cols,
                                   (float *)r->d_vals, r->d_counts);
        }
    }
    st = hipGetLastError() == hipSuccess ? SG_OK : SG_ERR_HIP;
    ctx->release(d_desc);
    if (st != SG_OK) {
        sg_topn_free(r);
        return st;
    }
    *out = r;
    return SG_OK;
}

extern "C" int sg_sp_matmul_topn_host(sg_ctx *ctx, int64_t n_left, int64_t n_right, int64_t n_cols,
                                      const int64_t *a_indptr, const int32_t *a_indices, const void *a_data,
                                      const int64_t *b_indptr, const int32_t *b_indices, const void *b_data,
                                      int32_t dtype, int32_t top_n, double threshold, int32_t sort,
                                      int32_t *out_cols, void *out_vals, int32_t *out_counts) {
    sg_csr *A = nullptr, *B = nullptr;
    sg_postings *P = nullptr;
    sg_topn *R = nullptr;
    int st = sg_csr_from_host(ctx, n_left, n_cols, a_indptr, a_indices, a_data, dtype, &A);
    if (st == SG_OK) st = sg_csr_from_host(ctx, n_right, n_cols, b_indptr, b_indices, b_data, dtype, &B);
    if (st == SG_OK) st = sg_postings_build(ctx, B, 0, &P);
    if (st == SG_OK) st = sg_spgemm_topn(ctx, A, P, top_n, threshold, sort, &R);
    if (st == SG_OK) st = sg_topn_to_host(ctx, R, out_cols, out_vals, out_counts);
    sg_topn_free(R);
    sg_postings_free(P);
    sg_csr_free(B);
    sg_csr_free(A);
    return st;
}
