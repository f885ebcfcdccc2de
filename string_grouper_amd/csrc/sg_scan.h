// Single-pass prefix sums (decoupled look-back) with the input and the output as FUNCTORS, so that a kernel that only
// produced a scan's input, or only consumed its output, can ride inside the scan (round 6: the grouping's "is this row its
// group's representative" and the position permutation's row lengths are computed by the scan that sums them).
// Included by the translation units that launch scans of their own; sg_api.hip holds the plain-array entry points.
#pragma once
#include "sg_internal.h"

// ------------------------------------------------------------------------------------ prefix sums
// ONE launch per scan (round 4; rounds 1-3: per-block totals -> recursive scan of the totals -> per-block apply, three to
// five launches and two extra passes over the totals -- a sixth of the step's 135 launches at 663 k were scans).
// Single pass with decoupled look-back: a tile (1024 threads x 8 items) sums its items, publishes {aggregate}, looks back
// over its predecessors' descriptors until it meets an inclusive prefix, publishes {inclusive prefix}, and writes its
// items.  HBM-bound: one read and one write per element.
//   * Tiles are numbered by a TICKET (atomic counter), not by blockIdx: a tile only ever waits for tiles with smaller
//     tickets, which have started -- forward progress whatever order the hardware schedules workgroups in.  The counter is
//     never reset: the host knows how many tickets every scan takes and passes the first one (`ticket_base`).
//   * A descriptor is ONE 64-bit word {epoch:16 | state:2 | value:46}, written and read with relaxed device-scope
//     atomics (the word carries everything: no ordering against other memory is needed).  The epoch names the scan, so the
//     descriptor array of the context is never cleared between scans (a clear would be the launch this saves); when the
//     16-bit epoch wraps -- every 65 535 scans -- it is cleared once.  Values are < 2^46 (7 * 10^13 entries).
//   * In-place (d_out == d_in) is fine: a tile holds its items in registers before it writes them.
#define SCAN_THREADS 1024      // (tiles of 8192 items: the look-back walks 64 predecessors per ~1.5 us round trip -- with
                               //  tiles of 2048 a scan of 5 M bins spent 50 us waiting for its 2 440 tiles' chain)
#define SCAN_ITEMS 8
#define SCAN_BLOCK (SCAN_THREADS * SCAN_ITEMS)
#define SCAN_VALUE_BITS 46
#define SCAN_STATE_AGGREGATE 1ull
#define SCAN_STATE_PREFIX 2ull

template <typename TO>
__device__ __forceinline__ TO block_exclusive_scan(TO v, TO *lds_wave_tot, TO *block_total) {
    // inclusive scan inside the 64-wide wave by shuffles, then across the waves through LDS
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    TO incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        TO up = __shfl_up(incl, d, 64);
        if (lane >= d) incl += up;
    }
    if (lane == 63) lds_wave_tot[wave] = incl;
    __syncthreads();
    TO wave_off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < SCAN_THREADS / 64; ++w) {
        TO t = lds_wave_tot[w];
        if (w < wave) wave_off += t;
        tot += t;
    }
    *block_total = tot;
    __syncthreads();
    return wave_off + incl - v;
}

__device__ __forceinline__ unsigned long long scan_desc(uint32_t epoch, unsigned long long state, unsigned long long value) {
    return ((unsigned long long)epoch << 48) | (state << SCAN_VALUE_BITS) | (value & ((1ull << SCAN_VALUE_BITS) - 1ull));
}

// Load: TO operator()(int64_t i) const -- item i (may have side effects of its own: it is called exactly once per i < n);
// Store: void operator()(int64_t i, TO exclusive_prefix) const.
template <typename TO, typename Load, typename Store>
__global__ void __launch_bounds__(SCAN_THREADS) scan_lookback_kernel(Load load, Store store, int64_t n, TO *d_total,
                                                                     unsigned long long *desc, uint32_t *ticket,
                                                                     uint32_t ticket_base, uint32_t epoch) {
    __shared__ TO wt[SCAN_THREADS / 64];
    __shared__ uint32_t s_tile;
    __shared__ unsigned long long s_excl;
    if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u) - ticket_base;
    __syncthreads();
    const uint32_t tile = s_tile;
    const int64_t base = (int64_t)tile * SCAN_BLOCK + (int64_t)threadIdx.x * SCAN_ITEMS;
    TO v[SCAN_ITEMS];
    TO s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        v[i] = (base + i < n) ? load(base + i) : (TO)0;
        s += v[i];
    }
    TO tot;
    TO run = block_exclusive_scan<TO>(s, wt, &tot);
    if (threadIdx.x < 64) {   // the tile's first wave publishes and looks back, 64 predecessors at a time
        const int lane = threadIdx.x;
        if (lane == 0)
            __hip_atomic_store(&desc[tile], scan_desc(epoch, tile == 0 ? SCAN_STATE_PREFIX : SCAN_STATE_AGGREGATE, (unsigned long long)tot),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned long long excl = 0;
        int64_t hi = (int64_t)tile;   // predecessors [.., hi) still to be added
        while (hi > 0) {
            const int64_t at = hi - 1 - lane;
            unsigned long long d = 0;
            bool ready = true;
            if (at >= 0) {
                d = __hip_atomic_load(&desc[at], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ready = (uint32_t)(d >> 48) == epoch && ((d >> SCAN_VALUE_BITS) & 3ull) != 0ull;
            }
            // lanes are ordered nearest predecessor first: use the ready descriptors up to (and including) the first
            // inclusive prefix, provided every one before it is ready; otherwise look again
            const uint64_t not_ready = __ballot(!ready);
            const uint64_t is_prefix = __ballot(at >= 0 && ready && ((d >> SCAN_VALUE_BITS) & 3ull) == SCAN_STATE_PREFIX);
            const int first_gap = not_ready ? __builtin_ctzll(not_ready) : 64;
            const int first_prefix = is_prefix ? __builtin_ctzll(is_prefix) : 64;
            const int usable = first_prefix < first_gap ? first_prefix + 1 : first_gap;   // lanes [0, usable)
            unsigned long long add = (lane < usable && at >= 0) ? (d & ((1ull << SCAN_VALUE_BITS) - 1ull)) : 0ull;
#pragma unroll
            for (int dd = 32; dd > 0; dd >>= 1) {
                const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)add, dd, 64), hh = (uint32_t)__shfl_xor((int)(uint32_t)(add >> 32), dd, 64);
                add += ((unsigned long long)hh << 32) | lo;
            }
            excl += add;
            if (first_prefix < first_gap) break;      // met an inclusive prefix: done
            hi -= usable;                               // (usable may be 0: spin on the same window)
            if (usable == 0) __builtin_amdgcn_s_sleep(1);
        }
        if (lane == 0) {
            if (tile != 0)
                __hip_atomic_store(&desc[tile], scan_desc(epoch, SCAN_STATE_PREFIX, excl + (unsigned long long)tot), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            s_excl = excl;
        }
    }
    __syncthreads();
    run += (TO)s_excl;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        if (base + i < n) store(base + i, run);
        run += v[i];
    }
    // the element one past the end receives the grand total (row-pointer convention)
    if (d_total && (int64_t)tile == (n - 1) / SCAN_BLOCK && threadIdx.x == SCAN_THREADS - 1) *d_total = run;
}

template <typename TO, typename Load, typename Store>
static int sg_scan_launch(sg_ctx *ctx, Load load, Store store, int64_t n, TO *d_total) {
    if (n <= 0) {
        if (d_total) SG_HIP_TRY(hipMemsetAsync(d_total, 0, sizeof(TO), ctx->stream));
        return SG_OK;
    }
    const int64_t nblocks = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
    SG_REQUIRE(nblocks < ((int64_t)1 << 31), "scan of more than 2^42 entries");
    // host threads that share a context launch on one stream: the tickets a launch will draw must be the ones the host
    // has told it about, in stream order
    std::lock_guard<std::mutex> scan_lock(ctx->scan_mu);
    if ((size_t)nblocks > ctx->scan_desc_cap) {   // grow the descriptor array (+ the ticket word behind it)
        size_t cap = ctx->scan_desc_cap ? ctx->scan_desc_cap : 4096;
        while (cap < (size_t)nblocks) cap *= 2;
        void *p = nullptr;
        SG_HIP_TRY(hipMalloc(&p, (cap + 2) * sizeof(unsigned long long)));
        if (hipMemsetAsync(p, 0, (cap + 2) * sizeof(unsigned long long), ctx->stream) != hipSuccess) {
            (void)hipFree(p);
            return SG_ERR_HIP;
        }
        if (ctx->d_scan_desc) {   // scans in flight on the stream still use the old array
            SG_HIP_TRY(hipStreamSynchronize(ctx->stream));
            (void)hipFree(ctx->d_scan_desc);
        }
        ctx->d_scan_desc = (unsigned long long *)p;
        ctx->scan_desc_cap = cap;
        ctx->scan_ticket_base = 0;
        ctx->scan_epoch = 0;
    }
    if (++ctx->scan_epoch > 0xFFFFu) {            // the 16-bit epoch wraps: stale descriptors could pass for fresh ones
        SG_HIP_TRY(hipMemsetAsync(ctx->d_scan_desc, 0, ctx->scan_desc_cap * sizeof(unsigned long long), ctx->stream));
        ctx->scan_epoch = 1;
    }
    uint32_t *ticket = reinterpret_cast<uint32_t *>(ctx->d_scan_desc + ctx->scan_desc_cap);
    hipLaunchKernelGGL((scan_lookback_kernel<TO, Load, Store>), dim3((unsigned)nblocks), dim3(SCAN_THREADS), 0, ctx->stream, load, store, n,
                       d_total, ctx->d_scan_desc, ticket, ctx->scan_ticket_base, ctx->scan_epoch);
    SG_HIP_TRY(hipGetLastError());                // (a launch that failed has drawn no tickets)
    ctx->scan_ticket_base += (uint32_t)nblocks;   // (mod 2^32, like the device counter)
    return SG_OK;
}


// plain arrays
template <typename TI, typename TO, bool NONZERO>
struct SgScanLoadArray {
    const TI *in;
    __device__ __forceinline__ TO operator()(int64_t i) const { return NONZERO ? (TO)(in[i] > 0 ? 1 : 0) : (TO)in[i]; }
};
template <typename TO>
struct SgScanStoreArray {
    TO *out;
    __device__ __forceinline__ void operator()(int64_t i, TO v) const { out[i] = v; }
};
template <typename TI, typename TO, bool NONZERO = false>
static int scan_impl(sg_ctx *ctx, const TI *d_in, TO *d_out, int64_t n, TO *d_total) {
    return sg_scan_launch<TO>(ctx, SgScanLoadArray<TI, TO, NONZERO>{d_in}, SgScanStoreArray<TO>{d_out}, n, d_total);
}
