// Device-side helpers shared by the kernels of the top-n multiply (sg_spgemm_topn.hip: exact kernel,
// zip; sg_spgemm_pruned.hip: pruned kernel).
#ifndef SG_K4_DEVICE_H
#define SG_K4_DEVICE_H
#include <math.h>

#include "sg_internal.h"

#define SG_TOPN_LANES 64

// Debug aid (-DSG_WATCHDOG): every data-dependent loop counts its iterations; an overrun records which
// loop it was in g_sg_watch and makes all loops wind down instead of hanging the GPU.
#ifdef SG_WATCHDOG
static __device__ int g_sg_watch[4];   // one per translation unit
#define SG_WD_DECL(c) int c = 0
#define SG_WD(c, limit, code)                                   \
    if (++(c) > (int)(limit) || ((volatile int *)g_sg_watch)[0]) { \
        if (((volatile int *)g_sg_watch)[0] == 0) {             \
            g_sg_watch[0] = (code);                             \
            g_sg_watch[1] = (int)(c);                           \
        }                                                       \
        break;                                                  \
    }
#ifndef SG_WATCH_NAME
#define SG_WATCH_NAME sg_debug_watch
#endif
extern "C" int SG_WATCH_NAME(int32_t *out4) {
    return hipMemcpyFromSymbol(out4, HIP_SYMBOL(g_sg_watch), 16) == hipSuccess ? 0 : 4;
}
#else
#define SG_WD_DECL(c)
#define SG_WD(c, limit, code)
#endif   // entries of the register-resident list = lanes of a wave

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
__device__ __forceinline__ T add_rn(T a, T b);
template <>
__device__ __forceinline__ float add_rn<float>(float a, float b) { return __fadd_rn(a, b); }
template <>
__device__ __forceinline__ double add_rn<double>(double a, double b) { return __dadd_rn(a, b); }

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
    // The same for a candidate that may already be in the list (same column => same pair => same score): entered once.
    // A repeat of an entry that has dropped off the end ranks behind all 64 entries and is refused like the first time.
    __device__ __forceinline__ void insert_unique(T ns, int nc, int lane) {
        if (__ballot(c == nc) != 0) return;
        insert(ns, nc, lane);
    }
};

// Two register lists behind each other: lane r holds the r-th and the (64 + r)-th best -- a row's result of 65 .. 128
// entries (the second pass of the self-join form, top_n above one register list).  What drops off the first list's end is
// the second list's new head.
template <typename T>
struct TopListWide {
    TopList<T> lo, hi;
    __device__ __forceinline__ void clear() {
        lo.clear();
        hi.clear();
    }
    __device__ __forceinline__ void insert(T ns, int nc, int lane) {
        const bool mine_first = (lo.s > ns) || (lo.s == ns && lo.c < nc);
        const int pos = __popcll(__ballot(mine_first));
        if (pos >= SG_TOPN_LANES) {
            hi.insert(ns, nc, lane);
            return;
        }
        const T es = wave_read<T>(lo.s, SG_TOPN_LANES - 1);
        const int ec = wave_read<int>(lo.c, SG_TOPN_LANES - 1);
        lo.insert(ns, nc, lane);
        if (ec != INT32_MAX) hi.insert(es, ec, lane);   // (ranks ahead of every entry of `hi`: position 0)
    }
    __device__ __forceinline__ void insert_unique(T ns, int nc, int lane) {
        if (__ballot(lo.c == nc || hi.c == nc) != 0) return;
        insert(ns, nc, lane);
    }
    __device__ __forceinline__ int count() const {
        return __popcll(__ballot(lo.c != INT32_MAX)) + __popcll(__ballot(hi.c != INT32_MAX));
    }
};

// Posting entry (written by K3): f32 -> packed {uint32 slot, float value}, one 8-byte load per lane;
// f64 -> slots[] (uint32) + vals[] (double).  "slot" is the BYTE offset of the entry's accumulator
// inside its column tile, (j mod TILE) * sizeof(T): the multiply never needs j itself -- the tile
// sweep recovers columns from positions -- so the address arithmetic is done once, in K3.
template <typename T>
struct Post;
template <>
struct Post<float> {
    typedef uint2 reg_t;
    static constexpr int STRIDE = 8;
    // address = kernel-constant base (SGPR pair) + 32-bit byte offset per lane: no 64-bit arithmetic at all
    static __device__ __forceinline__ reg_t load(const char *vals, const char *, uint32_t seg_lo, uint32_t entry) {
        return *reinterpret_cast<const uint2 *>(vals + ((seg_lo + entry) << 3));   // 32-bit offset: < 2^29 entries
    }
    static __device__ __forceinline__ uint32_t slot(const reg_t &r) { return r.x; }
    static __device__ __forceinline__ float val(const reg_t &r) { return __uint_as_float(r.y); }
};
template <>
struct Post<double> {
    struct reg_t {
        uint32_t j;
        double v;
    };
    static constexpr int STRIDE = 8;
    static __device__ __forceinline__ reg_t load(const char *vals, const char *slots, uint32_t seg_lo, uint32_t entry) {
        reg_t r;
        r.j = *reinterpret_cast<const uint32_t *>(slots + ((seg_lo + entry) << 2));
        r.v = *reinterpret_cast<const double *>(vals + ((seg_lo + entry) << 3));
        return r;
    }
    static __device__ __forceinline__ uint32_t slot(const reg_t &r) { return r.j; }
    static __device__ __forceinline__ double val(const reg_t &r) { return r.v; }
};


// Next left row for this wave: one global atomic by lane 0, broadcast.  The result is made
// explicitly wave-uniform so that everything derived from it stays in SGPRs / uniform branches.
__device__ __forceinline__ uint32_t next_row(uint32_t *row_counter, int lane) {
    uint32_t r = 0;
    if (lane == 0) r = __hip_atomic_fetch_add(row_counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)r);
}

#endif   // SG_K4_DEVICE_H
