// K4p -- the top-n multiply with exact prefix-filter pruning (cosine data: non-negative values, row
// norms <= 1, i.e. what the TF-IDF vectoriser produces; anything else runs the exact kernel K4).
//
// Same contract as K4 (sparse_dot_topn.sp_matmul_topn as called from string_grouper/string_grouper.py
// :725-732 of the reference) and bit-identical results; it only avoids work that provably cannot
// produce a score above the threshold.  The survey lists this as the remedy for the skew of name data
// (hard part 3): a handful of n-grams ('inc', 'llc', ' co') are present in a fifth
// of all rows and generate most of the intermediate products, yet carry almost no weight.
//
// For left row i (non-zeros a_k) the terms are split into a suffix S -- the most frequent terms, all of
// them "frequent" (list length >= freq_min), as many as fit under ||a_S|| * max_j ||b_j|| <= beta =
// threshold - delta -- and a prefix P (the rest: the rare terms).  With f_j = ||b_j restricted to the
// frequent terms||, Cauchy-Schwarz gives
//        score(i, j) = sum_{k in P} a_k b_jk + sum_{k in S} a_k b_jk <= p_ij + ||a_S|| f_j <= p_ij + beta
// so (1) a right row j that shares no term of P with row i cannot reach the threshold -- only the
// posting lists of P are streamed (4-25x fewer entries than all lists), and (2) of the rows that do,
// only those with p_ij > threshold - ||a_S|| f_j need their score computed.  Those "survivors" are
// scored EXACTLY -- row j of B is walked in ascending k, every term looked up in row i, product and sum
// rounded separately: the reference's arithmetic and summation order -- then the strict threshold,
// the canonical order (score desc, column asc) and the top-n cut are applied as in K4.
//
// p_ij is only a filter: any upper bound will do, in any summation order.  K3 therefore writes, next
// to the postings proper, 4-byte "filter postings" {accumulator address + half, bq, fq: 8} with the
// value b and the norm f_j quantised UPWARDS, and the kernel accumulates upper bounds of a * b in
// 16-bit fixed point (scale 2^15) with LDS integer atomics, all in integer arithmetic:
//        x    = (CA * (bq << AB)) >> 32   >=  a * b * 2^15 - 1    CA = ceil(a * norm_b * 2^15 / bq_max * 2^(32 - AB))
//        tq   = (T0 - C1 * fq) >> 8       <=  (thr - 1e-5) * 2^15 - 2 - ||a_S|| f_j 2^15 - |P|
// (the |P| pays for the "- 1" of every add: a column receives at most one posting per term of P) and column j
// survives when its accumulator reaches tq.  (These are the tile-by-tile form's formulas; the stream form, the default
// since round 3, has the same bound in sixteen more bits and its postings' fields cut for the instructions that read
// them -- round 4: "stream form" in the kernel, emit_posting in sg_postings.hip, tests/test_prune_model.py.)
// Because atomics tolerate collisions, every
// lane of the wave carries a posting of whatever term: lanes are dealt to the terms of P in
// proportion to their list lengths, lane (term g, u of G_g) owns entries lo_g + u, lo_g + u + G_g, ...
// of its term's list.  ds_add_rtn_u32 returns the previous value, so the lane whose add takes an
// accumulator across its column's survivor threshold knows it (values only grow) and appends the column
// to the wave's survivor list -- no sweep of the tile.  Measured on MI355X: 7.6-13 cycles per 64-lane
// ds_add(_rtn)_u32 per CU against 16.5 for a plain read-add-write and 212 for ds_add_f32
// (profiles/r01_lds_atomic_microbench.log).
//
// The column-tile loop (round 2).  Round 1's loop spent ~150 VALU + ~67 SALU wave instructions per
// (row, tile) for ~1.5 postings per lane and ran at 94 % VALU-pipe utilisation (profiles/r01_profile_k4p.log):
// float conversions for x and tq, a select + shift + compare per speculative load, register rotation of the
// prefetched batches.  Now: the loop is unrolled four tiles deep with four statically named batches (the
// batch of tile t + 3 is issued while tile t is applied -- no register moves, so no waits on loads in
// flight); a batch is ONE unconditional 16-byte load -- lane u of the g lanes of a term owns entries 4u .. 4u + 3 of
// the term's segment in every tile (K3 leaves slack behind the postings) -- plus ONE signed remainder `rem` = bytes
// of the segment at and after the lane's first entry, slot e is valid iff 4e < rem; the segment ends come four
// tiles per 16-byte load from the padded table K3 writes; x and tq are 24-bit integer multiplies (v_mul_hi_u32_u24,
// v_mul_i32_i24 with a byte select).  ~12 VALU per slot, four slots per tile.
//
// Self-join ("symmetric") mode: score(i, j) and score(j, i) are the same float (same products, same
// ascending-k order), so when A is the matrix the postings were built from, row i only walks the tiles
// up to its own column and scores the pairs j <= i: its own matches go into its register top-n list and out to its
// result row, the MIRRORED pairs (that i matches j < i) into a chunked global list, and a second pass merges them
// into the rows they point at (pairs_fill / pairs_select).  Half the (row, tile) visits, half the exact scorings,
// identical result.
//
// Position space: "column", "tile" and "row i of the self-join" mean POSITIONS of a fixed permutation of the right-hand
// rows (sg_postings.hip: a sorted name list piles a row's candidates into a few tiles otherwise); what leaves the
// kernel -- the result row, the columns kept, the pairs -- names rows again ({pointer, row} per position rides with
// the packed rows; equal scores are ordered and cut by the ORIGINAL column).
//
// One 64-lane wave per left row, single-wave workgroups, persistent waves fed by a global row
// counter, as K4.  LDS per wave: the tile's 4096 u16 accumulators (two per word), row i as a term ->
// value hash (for the exact scoring) and the survivor buffer: 10 KiB -> 16 waves per CU.
// Rows with 65 .. 128 non-zeros are appended to a list and taken by a second launch of the same kernel that stages two
// terms per lane (WIDE); what that one cannot take either (more than 128 non-zeros, more than 64 prefix terms) goes to
// K4 (in symmetric mode: to K4's self-join launch, spgemm_topn_selfjoin_rows_kernel, inside the same pass).
#define SG_WATCH_NAME sg_debug_watch_pruned
#include "sg_k4_device.h"

#ifdef SG_DEBUG_WAVE_TIMES   // (scripts/build_variant.sh times -DSG_DEBUG_WAVE_TIMES; scripts/wave_times_probe.py)
// per wave of the last launch over rows ([0, 8192)) and over parts ([8192, 16384)): {start, end} on the 100 MHz clock,
// the wave's slowest row: its duration, row | pairs scored << 32
__device__ unsigned long long sg_debug_wave_times[4 * 16384];
extern "C" int sg_debug_read_wave_times(unsigned long long *out, int clear) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(sg_debug_wave_times), sizeof(sg_debug_wave_times)) != hipSuccess) return 1;
    if (clear) {
        static unsigned long long zero[4 * 16384];
        if (hipMemcpyToSymbol(HIP_SYMBOL(sg_debug_wave_times), zero, sizeof(zero)) != hipSuccess) return 1;
    }
    return 0;
}
#endif

// (SG_PAIR_CHUNK, the entries of a chunk of the symmetric mode's pair list: sg_internal.h)
// Waves per SIMD the f64 stream kernels are built for (A/B knob of the build: scripts/build_variant.sh).  Round 3 built
// them for three (at 128 registers the visit ends were spilled inside the round loop); with round 4's loop they fit 128
// registers with five spilled around the row's set-up and none in the loop, and LDS (10.5 KiB) allows 15 waves per CU:
// 7.0 -> 6.5 ms at 663 k.
#ifndef SG_F64_STREAM_WAVES
#define SG_F64_STREAM_WAVES 4
#endif
#define SG_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// Probe builds (WRONG results, timing only; a library per probe: scripts/build_variant.sh <name> -D<macro>, A/B through
// SG_HIP_LIB): the product kernel carries two, both in the survivor routine -- SG_PROBE_NO_DRAIN (the routine returns at
// once: what candidates cost in all) and SG_PROBE_Q8_ONLY (second filter without the exact scoring behind it) -- and the
// per-wave clocks of SG_DEBUG_WAVE_TIMES (right results).  The fourteen probes of rounds 2 - 4 (no loads, no LDS, no collect,
// plain loads, round counts ...) did their work and are gone from the source: profiles/HISTORY.md has what they measured,
// commit fd184d0 the code.
// Parts of a row in the launch over parts (stream + self-join form).  Sixteen through most of round 4 -- until per-wave
// clocks of an eighth share of the 5 M job (profiles/r04_final_wave_times.log) showed the launch over parts ending 4.2 ms
// after its median wave, on FOUR waves: the heaviest rows of that job are 10 - 50 ms of work each (153 visits), and a
// sixteenth of that is still milliseconds.  Parts that hold no visit (a row of fewer visits than parts) cost an iteration.
#define SG_ROW_PARTS_LOG2 6
#define SG_ROW_PARTS 64u
#define SG_SURV_CAP 128   // survivors buffered per wave (scored 64 at a time as soon as 64 are there)

// lane mask of a predicate as a wave-uniform scalar (s_and of the compare result, no VALU round trip)
__device__ __forceinline__ uint64_t ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }

template <typename T>
__device__ __forceinline__ T wave_shfl(T v, int src) {
    return __shfl(v, src, 64);
}

// Row i of A is staged in LDS as a 128-slot open-addressing hash (term -> value) so that a lane can walk
// row j of B with INDEPENDENT loads -- eight entries in flight per round, not one dependent load per merge
// step -- and look every term up in one or two LDS reads.
#define SG_HASH_SLOTS 128
// The hash names a BUCKET of four consecutive slots (its first slot: a multiple of four); a term sits in the first slot
// that was free at or behind it when the row was staged (compare-and-swap, linear from there, wrapping).  So a bucket with
// a free slot holds every term that hashes to it, and ONE 16-byte LDS read -- the bucket's four keys -- settles a lookup
// unless the bucket is full and does not hold the term (rare at <= 64 terms in 32 buckets: then the walk goes on slot by
// slot).  The second filter looks four entries up side by side this way (q8_unit): two LDS round trips per four entries;
// slot-by-slot probing took seven, and those round trips were most of what the survivor routine cost.
// (A 24-bit multiply: full rate, where the 32-bit one takes four passes.)
#ifdef SG_HASH_FINE   // (probe: a home SLOT instead of a home bucket; only with SG_Q8=0 -- the second filter reads buckets)
__device__ __forceinline__ uint32_t term_hash(int k) { return (__umul24((uint32_t)k & 0xffffffu, 0x9E3779u) >> 15) & 127u; }
#else
__device__ __forceinline__ uint32_t term_hash(int k) { return (__umul24((uint32_t)k & 0xffffffu, 0x9E3779u) >> 15) & 124u; }
#endif

// An object of the wave's LDS by its byte address.  The kernel's dynamic LDS is its only LDS and starts at address 0
// (the accumulator tile first: a posting's address field IS an LDS address).  In the kernel `smem + x` is as good; in the
// routines the kernel CALLS it is not: there the compiler finds the array through a table in memory
// (llvm.amdgcn.dynlds.offset.table) with a scalar load and a wait in front of EVERY access -- eight per round of
// entries in the exact scoring.
template <typename P>
__device__ __forceinline__ P *lds_object(uint32_t byte_addr) {
    typedef __attribute__((address_space(3))) P lds_t;
    return (P *)(lds_t *)(uintptr_t)byte_addr;
}

// A pointer into device memory as one of the GLOBAL address space.  Pointers that reach a routine through a struct are
// generic to the compiler: flat_load instead of global_load -- counted by the LDS counter as well, so that every wait for
// an LDS read also waits for the loads in flight.
template <typename P>
__device__ __forceinline__ const __attribute__((address_space(1))) P *global_ptr(const void *p) {
    return (const __attribute__((address_space(1))) P *)(uintptr_t)p;
}
// (HIP's uint4 / uint2 are classes whose copy constructors take generic references: the loads go through the native vectors)
typedef uint32_t sg_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t sg_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint4 load_global_u4(const void *base, size_t idx) {
    const sg_u32x4 v = global_ptr<sg_u32x4>(base)[idx];
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint2 load_global_u2(const void *base, size_t idx) {
    const sg_u32x2 v = global_ptr<sg_u32x2>(base)[idx];
    return make_uint2(v.x, v.y);
}

// A small struct of launch constants (SgScoreCtx, SgPairSink) read through the SCALAR cache.  The routines below are real
// calls and get the struct's address in a vector register: every field was a vector load of its own with a full wait
// behind it -- ten dependent round trips per call of drain_survivors before the first entry of a candidate was
// requested.  The address is the same in every lane and nothing writes the struct while the kernel runs: constant
// address space, s_load.
template <typename S>
__device__ __forceinline__ S load_launch_constants(const S *p) {
    static_assert(sizeof(S) % 4 == 0, "dwords");
    const uint64_t a = (uint64_t)(uintptr_t)p;
    const uint64_t u = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32)) << 32) |
                       (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a);
    typedef const __attribute__((address_space(4))) uint32_t *const_words;
    const const_words w = (const_words)(uintptr_t)u;
    union {
        S s;
        uint32_t d[sizeof(S) / 4];
    } out = {};
#pragma unroll
    for (unsigned i = 0; i < sizeof(S) / 4; ++i) out.d[i] = w[i];
    return out.s;
}

// Packed rows of B (sg_postings.hip, fwd_pack): eight entries per round, as 16-byte loads.
template <typename T>
struct FwdRound;
template <>
struct FwdRound<float> {   // 4 loads x 2 entries; q is even
    int k[8];
    float v[8];
    __device__ __forceinline__ void load(const void *fwd, uint32_t q, uint32_t last) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint4 w = load_global_u4(fwd, min((q >> 1) + e, last >> 1));
            k[2 * e] = (int)w.x;
            v[2 * e] = __uint_as_float(w.y);
            k[2 * e + 1] = (int)w.z;
            v[2 * e + 1] = __uint_as_float(w.w);
        }
    }
};
template <>
struct FwdRound<double> {   // 8 loads x 1 entry
    int k[8];
    double v[8];
    __device__ __forceinline__ void load(const void *fwd, uint32_t q, uint32_t last) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const uint4 w = load_global_u4(fwd, min(q + e, last));
            k[e] = (int)w.x;
            v[e] = __longlong_as_double((long long)(((unsigned long long)w.w << 32) | w.z));
        }
    }
};

// Value of term `key` in row i as staged in LDS.  Rows of up to 64 terms: a 128-slot open-addressing hash (hk, ha).
// Wide rows (65 .. 128 terms): the row's sorted terms and values (hk = terms, ha = values), binary search -- a full
// 128-slot hash has no empty slot to stop a probe and a larger one does not fit beside the accumulator tile.
template <typename T, bool WIDE>
__device__ __forceinline__ T row_value(const int *hk, const T *ha, int key, int nnz) {
    if (WIDE) {
        int lo = 0, hi = nnz;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (hk[mid] < key) lo = mid + 1;
            else hi = mid;
        }
        return (lo < nnz && hk[lo] == key) ? ha[lo] : (T)0;
    }
    uint32_t h = term_hash(key);
    int got = hk[h];
    SG_WD_DECL(wd_p);
    while (got != key && got != -1) {
        SG_WD(wd_p, SG_HASH_SLOTS + 2, 23)
        h = (h + 1) & (SG_HASH_SLOTS - 1);
        got = hk[h];
    }
    return got == key ? ha[h] : (T)0;
}

// Exact score of (row i of A, row j of B) for the lanes with j >= 0: the products of the shared terms are
// added in ascending k (B's rows are sorted), product and sum rounded separately -- the reference's
// arithmetic.  A term row i does not have contributes a * b with a = 0: sum + 0 == sum exactly (all
// values are non-negative), so absent terms and the padding of a round need no branch.
template <typename T, bool WIDE>
__device__ __forceinline__ T exact_score_packed(bool have, uint32_t pb, uint32_t pe, const int *hk, const T *ha, int nnz,
                                                const void *__restrict__ fwd) {
    T sum = (T)0;
    if (have) {
        // A call is a chain of dependent misses -- (the row's pointer, then) its entries eight at a time: four in a row for
        // a row of 19 entries (the mean of a name list), ~10 us per call of 64 candidates, a fifth of the kernel.  The
        // rounds behind the first are therefore TOUCHED (one dword each, three registers) together with the first round's
        // loads: when their turn comes they arrive from the cache.  (Two rounds in flight by name instead -- 32 registers
        // -- spilled in every kernel of the file.)
        constexpr uint32_t DW = sizeof(T) == 4 ? 2u : 4u;   // dwords per entry
        const auto *fw = global_ptr<uint32_t>(fwd);
        const uint32_t q0 = pb & ~1u;
        uint32_t t1 = 0, t2 = 0, t3 = 0;
        if (q0 + 8u < pe) t1 = fw[(q0 + 8u) * DW];
        if (q0 + 16u < pe) t2 = fw[(q0 + 16u) * DW];
        if (q0 + 24u < pe) t3 = fw[(q0 + 24u) * DW];
        SG_WD_DECL(wd_v);
        for (uint32_t q = q0; q < pe; q += 8) {
            SG_WD(wd_v, 1 << 20, 21)
            FwdRound<T> r;
            r.load(fwd, q, pe - 1u);
            // (the lookups one after the other: batched like the second filter's -- row_values -- they cost the routine 15 to 30
            //  registers more, which count against the round loop's kernel, and bought nothing at 663 k: 4.80 against 4.70 ms)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                T a = (T)0;
                if (q + e >= pb && q + e < pe) a = row_value<T, WIDE>(hk, ha, r.k[e], nnz);
                sum = add_rn<T>(sum, mul_rn<T>(a, r.v[e]));
            }
        }
        asm volatile("" ::"v"(t1), "v"(t2), "v"(t3));
    }
    return sum;
}

template <typename T, bool WIDE>
__device__ __forceinline__ T exact_score(int j, const int *hk, const T *ha, int nnz, const uint32_t *__restrict__ fwd_ptr,
                                         const void *__restrict__ fwd, int &row_of_j) {
    row_of_j = j;
    uint32_t pb = 0, pe = 0;
    if (j >= 0) {
        // {first entry, the row's own index} of positions j and j + 1: sixteen contiguous bytes
        const uint2 m0 = load_global_u2(fwd_ptr, (size_t)j), m1 = load_global_u2(fwd_ptr, (size_t)j + 1);
        pb = m0.x;
        pe = m1.x;
        row_of_j = (int)m0.y;
    }
    return exact_score_packed<T, WIDE>(j >= 0, pb, pe, hk, ha, nnz, fwd);
}

// SECOND FILTER (round 5).  The first filter's candidates are mostly false alarms -- sums of eight columns folded onto one
// accumulator, a bound that knows only the norm of a candidate's frequent part: 31.7 M pairs scored for 1.2 M above the
// threshold at 663 k -- and every one of them cost the exact scoring a pointer line plus two or three lines of a ~150-byte
// packed row, fetched one behind the other.  K3 therefore keeps an 8-BIT COPY of every right-hand row at a fixed stride
// (SgScoreCtx::q8, sg_internal.h; sg_postings.hip: q8_write_unit): one aligned line (two for rows of more than 29
// entries), addressed by the candidate's position alone.  Its values are rounded UP, so
//        U = sum_k a_k * bq_k   (over the candidate's entries; a_k = 0 for terms row i does not have)
// satisfies  score(i, j) <= U * norm_up / 255,  and a candidate with  U < (threshold - 3e-5) * 255 / norm_up  cannot
// reach the threshold: the exact kernel's float score obeys score~ <= score + 1e-5 (the first filter's own allowance),
// the float evaluation of U loses at most n * 2^-24 of it (8e-6 at the threshold for 128 terms), the conversion of a
// double a_k to float 6e-8 of it, q8_scale is rounded down -- 3e-5 covers all of it.  tests/test_prune_model.py restates
// both sides (no false negative for scores at the threshold +- 1 ulp, hubs, rows beyond the copy's 60 entries, f64).
// A row without a copy (more than SG_Q8_MAX_ENTRIES = 60 entries) always passes.  The header also holds what the exact scoring needs of the
// row -- first packed entry, number of entries, the row's own index -- so a candidate that passes costs no pointer fetch.
// N terms looked up in row i's hash side by side: the N buckets' keys in one go (16 bytes each), the (rare) walks behind
// full buckets, then the N values; out[q] = a_k, or 0 for a term row i does not have.  Two LDS round trips per N lookups
// where one lookup after the other takes two each, or more.  (N = 4 in the second filter.)
template <typename T, int N>
__device__ __forceinline__ void row_values(const int *hk, const T *ha, const int (&t)[N], T (&out)[N]) {
    uint32_t h[N];
    bool found[N];
    {
        int4 K[N];
#pragma unroll
        for (int q = 0; q < N; ++q) h[q] = term_hash(t[q]);
#pragma unroll
        for (int q = 0; q < N; ++q) K[q] = *reinterpret_cast<const int4 *>(hk + h[q]);
#pragma unroll
        for (int q = 0; q < N; ++q) {
            const bool m1 = K[q].y == t[q], m2 = K[q].z == t[q], m3 = K[q].w == t[q];
            found[q] = K[q].x == t[q] || m1 || m2 || m3;
            bool walking = !found[q] && K[q].w != -1;   // the bucket is full and does not hold the term: on, slot by slot
            h[q] += m1 ? 1u : (m2 ? 2u : (m3 ? 3u : 0u));
            if (__ballot(walking) != 0) {
                // (a loop the whole wave leaves together, every lane's state in registers of its own: hipcc 7.2 compiled the
                //  per-lane `while (k != t && k != -1)` of this walk with `found = k == t` taken from the LAST trip's compare
                //  mask -- lanes that had left the loop earlier lost their hit, and a row with five terms in one bucket lost
                //  four matches: scripts/q8_debug.py)
                uint32_t g = (h[q] + 4u) & (SG_HASH_SLOTS - 1);
                int k = -1;
                SG_WD_DECL(wd_p);
                while (__ballot(walking) != 0) {
                    SG_WD(wd_p, SG_HASH_SLOTS + 2, 26)
                    if (walking) {
                        k = hk[g];
                        walking = k != t[q] && k != -1;
                        if (walking) g = (g + 1u) & (SG_HASH_SLOTS - 1);
                    }
                }
                asm volatile("" : "+v"(k), "+v"(g));   // (the hit is read off the key, in a register, after the loop)
                if (!found[q] && K[q].w != -1) {
                    found[q] = k == t[q];
                    h[q] = g;
                }
            }
        }
    }
    T v[N];
#pragma unroll
    for (int q = 0; q < N; ++q) v[q] = ha[h[q]];   // (the slot of a term that is absent holds whatever: selected away)
#pragma unroll
    for (int q = 0; q < N; ++q) out[q] = found[q] ? v[q] : (T)0;
}

template <typename T, bool WIDE>
__device__ __forceinline__ float q8_unit(uint4 w, const int *hk, const T *ha, int nnz, float ub) {
    const uint32_t e[4] = {w.x, w.y, w.z, w.w};   // (an entry behind the row's last is 0: term 0 times bq = 0)
    if (WIDE) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            ub = __builtin_fmaf((float)row_value<T, true>(hk, ha, (int)(e[q] >> 8), nnz), (float)(e[q] & 255u), ub);
        return ub;
    }
    const int t[4] = {(int)(e[0] >> 8), (int)(e[1] >> 8), (int)(e[2] >> 8), (int)(e[3] >> 8)};
    T a[4];
    row_values<T, 4>(hk, ha, t, a);
#pragma unroll
    for (int q = 0; q < 4; ++q) ub = __builtin_fmaf((float)a[q], (float)(e[q] & 255u), ub);
    return ub;
}

template <typename T, bool WIDE>
__device__ __forceinline__ bool q8_passes(int j, const int *hk, const T *ha, int nnz, const uint4 *__restrict__ q8, float bar,
                                          uint32_t &pb, uint32_t &pe, int &row_of_j) {
    const bool have = j >= 0;
    const size_t rec = (size_t)(have ? j : 0) * (SG_Q8_STRIDE / 16u);   // (in 16-byte units)
    uint4 cur = make_uint4(0u, 0u, 0x80000000u, 0u), nxt = make_uint4(0u, 0u, 0u, 0u);
    if (have) {   // header and first entries together: the same 128-byte line
        cur = load_global_u4(q8, rec);
        nxt = load_global_u4(q8, rec + 1);
    }
    const uint32_t n_j = cur.z & 0x7fffffffu;
    const bool whole = (cur.z >> 31) == 0u;
    pb = cur.x;
    pe = cur.x + n_j;
    row_of_j = have ? (int)cur.y : j;
    const uint32_t units = whole ? (n_j + 7u) >> 2 : 1u;   // 16-byte units of the record in use (unit 0: the header)
    float ub = 0.f;
    SG_WD_DECL(wd_q);
    for (uint32_t u = 1u;; ++u) {   // the unit behind the one being worked off is on its way (same line: from the cache)
        SG_WD(wd_q, 20, 25)
        if (__ballot(u < units) == 0) break;
        cur = nxt;
        if (u + 1u < units) nxt = load_global_u4(q8, rec + u + 1u);
        if (u < units) ub = q8_unit<T, WIDE>(cur, hk, ha, nnz, ub);
    }
    return have && (!whole || ub >= bar);
}

// The same score from the ROW BLOCKS (SgScoreCtx::blk): rows at a fixed stride, 128-byte aligned, header first.  No row
// pointer to fetch before the row itself, every 128-byte line of a row is one aligned request, and the line behind the one
// being worked off is touched (one dword) as soon as the header says the row reaches it, so that its 64-byte units arrive
// from the cache.  The packed rows cost a dependent miss per step -- pointer, eight entries, the next eight ... -- and are
// gathered in unaligned pieces: 57.9 GB of memory-side traffic per launch at 663 k against 45.2 GB with the blocks
// (profiles/r03_sessionJ_traffic.log).  Deliberately small: ONE buffer of two 16-byte registers (32 bytes, a quarter of a
// line) -- this routine is called from inside the round loop, and every register it uses is one the loop cannot keep a
// value in across its call sites: with four registers per unit it needed 67 and the loop's kernel lost 0.3 ms to spills,
// with two line buffers 120 (the loop itself spilled).
// Same arithmetic: ascending entries, product and sum rounded separately, a = 0 for terms row i does not have.
template <typename T, bool WIDE>
__device__ __forceinline__ T exact_score_blocks(int j, const int *hk, const T *ha, int nnz, const SgScoreCtx *__restrict__ sc,
                                                int &row_of_j) {
    constexpr int ES = sizeof(T) == 4 ? 8 : 16;
    constexpr int EPU = 32 / ES;   // entries of a 32-byte unit (two 16-byte loads): 4 / 2; four units per 128-byte line
    T sum = (T)0;
    row_of_j = j;
    const bool have = j >= 0;
    const uint4 *base = reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(sc->blk) +
                                                        (size_t)(have ? j : 0) * (size_t)sc->blk_bytes);
    uint4 U[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) U[e] = make_uint4(0u, 0u, 0u, 0u);
    if (have) {
#pragma unroll
        for (int e = 0; e < 2; ++e) U[e] = base[e];
    }
    const int nnz_j = have ? (int)U[0].y : -1;    // header: {the row's own index, its entries}; entries 1 .. nnz_j follow
    if (have) row_of_j = (int)U[0].x;
    uint32_t touched = 0;
    auto use = [&](int idx, uint32_t term, T val) {
        const bool valid = idx >= 1 && idx <= nnz_j;
        T a = (T)0;
        if (valid) a = row_value<T, WIDE>(hk, ha, (int)term, nnz);
        sum = add_rn<T>(sum, mul_rn<T>(a, valid ? val : (T)0));   // (a unit that was not loaded holds the previous one)
    };
    SG_WD_DECL(wd_b);
    for (int u = 0;; ++u) {
        SG_WD(wd_b, 64, 24)
        if ((u & 3) == 0 && have && (u + 4) * EPU <= nnz_j)   // first unit of a line: touch the next line
            touched |= reinterpret_cast<const uint32_t *>(base)[(u + 4) * 8];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            if (sizeof(T) == 4) {
                use(u * EPU + 2 * e, U[e].x, (T)__uint_as_float(U[e].y));
                use(u * EPU + 2 * e + 1, U[e].z, (T)__uint_as_float(U[e].w));
            } else {
                use(u * EPU + e, U[e].x, (T)__longlong_as_double((long long)(((unsigned long long)U[e].w << 32) | U[e].z)));
            }
        }
        const bool more = have && (u + 1) * EPU <= nnz_j;   // the row reaches unit u + 1
        if (__ballot(more) == 0) break;
        if (more) {
#pragma unroll
            for (int e = 0; e < 2; ++e) U[e] = base[(u + 1) * 2 + e];
        }
    }
    asm volatile("" ::"v"(touched));
    return sum;
}

// Scores the columns in surv[0 .. min(n_surv, 64)) and moves the rest of the buffer to the front.  Deliberately
// not inlined: the tile loop has sixteen unrolled rounds and must not carry sixteen copies of this.
// (LDS objects are addressed through the kernel's own shared array so that they stay ds_* accesses.)
template <typename T, bool SYM, int TILE_LOG2, bool WIDE, bool UNIQ>
__device__ __noinline__ TopList<T> drain_survivors(int nnz, T thr, uint32_t row,
                                                   const SgScoreCtx *__restrict__ sc /* packed rows of B, position -> row */,
                                                   const SgPairSink *__restrict__ pairs /* SYM: the pair list */,
                                                   TopList<T> top, uint32_t n_surv) {
    constexpr int TILE = 1 << TILE_LOG2;
    const int *hk = lds_object<int>(TILE * 2);
    const T *ha = lds_object<T>(TILE * 2 + 512);
    int *surv = lds_object<int>(TILE * 2 + 512 + 1024);
    const int lane = threadIdx.x;
#ifdef SG_PROBE_NO_DRAIN   // timing probe (wrong results): what the whole routine costs
    if (n_surv > 64) {
        const uint32_t rem = n_surv - 64;
        const int keepv = (uint32_t)lane < rem ? surv[64 + lane] : 0;
        __builtin_amdgcn_wave_barrier();
        if ((uint32_t)lane < rem) surv[lane] = keepv;
    }
    return top;
#endif
    // (bit 31 of `row`: the caller has the row's match with itself already -- self-join, stream form -- and the diagonal
    //  among the candidates is left alone)
    // (bit 30: top_n above one register list -- the row's own matches j < i go to the pair list as well, addressed to the row
    //  itself, and the second pass, whose lists hold 128, selects: pairs_select_kernel.  Positions stay below 2^30.)
    const bool own_diag = (row >> 31) != 0u;
    const bool both = SYM && ((row >> 30) & 1u) != 0u;
    row &= 0x3fffffffu;
    int j = (uint32_t)lane < n_surv ? surv[lane] : -1;   // a POSITION: the index is built over a permutation of B's rows
    if (own_diag && (uint32_t)j == row) j = -1;
    const SgScoreCtx scv = load_launch_constants(sc);
    // what leaves the kernel is the row itself: the top list orders equal scores by the ORIGINAL column, the pairs name rows
    int jo;
    T sum;
    if (UNIQ && scv.q8 != nullptr) {   // (stream form) second filter first; the exact scoring only for what it lets through
        uint32_t pb, pe;
        const float bar = ((float)thr - 3e-5f) * scv.q8_scale;
#ifdef SG_PROBE_Q8_ONLY    // timing probe (wrong results): the second filter without the exact scoring behind it
        const bool pass = q8_passes<T, WIDE>(j, hk, ha, nnz, scv.q8, bar, pb, pe, jo) && (pb == 0xFFFFFFFFu);
#else
        const bool pass = q8_passes<T, WIDE>(j, hk, ha, nnz, scv.q8, bar, pb, pe, jo);
#endif
        const uint64_t pm = __ballot(pass);
        // pairs scored exactly, per wave, in the word of the survivor buffer in front of the pair list's (the stream
        // form buffers 126 columns at most); added to the launch's statistics when the wave ends
        if (lane == 0) surv[SG_SURV_CAP - 2] += (int)__popcll(pm);
        sum = pm ? exact_score_packed<T, WIDE>(pass, pb, pe, hk, ha, nnz, scv.fwd) : (T)0;
    } else {
        sum = scv.blk ? exact_score_blocks<T, WIDE>(j, hk, ha, nnz, &scv, jo)
                      : exact_score<T, WIDE>(j, hk, ha, nnz, scv.fwd_ptr, scv.fwd, jo);
        if (UNIQ && lane == 0) surv[SG_SURV_CAP - 2] += (int)min(n_surv, 64u);
    }
    uint64_t hm = __ballot(j >= 0 && sum > thr);
    if (SYM) {
        // row i keeps its own matches j <= i in its top list like the one-sided form; what row j < i has to learn -- that
        // i matches it -- goes to the pair list (pass 2 merges it into row j's list).  The diagonal is nobody's mirror.
        const uint64_t mm = hm & __ballot((uint32_t)j != row);
        if (mm) {
            // Every wave appends to a chunk of the pair list that it owns (SG_PAIR_CHUNK entries, taken from a global counter:
            // one returning atomic per chunk, not per row -- a returning atomic per row on one shared counter cost the
            // kernel a fifth of its time).  Where it stands -- chunk << 9 | entries used -- lives in LDS, in the last word
            // of the survivor buffer (which holds at most 127 columns).
            uint32_t pos = (uint32_t)surv[SG_SURV_CAP - 1];
            const uint32_t n_hit = (uint32_t)__popcll(mm) << (both ? 1 : 0);
            const SgPairSink pv = load_launch_constants(pairs);
            uint32_t *const pair_i = pv.d_i, *const pair_j = pv.d_j, *const pair_row_count = pv.d_row_count;
            T *const pair_s = reinterpret_cast<T *>(pv.d_s);
            const uint32_t pair_chunks = pv.chunks;
            // the left row's own name (self-join: A is B), the same for every lane
            const uint32_t row_name = (uint32_t)__builtin_amdgcn_readfirstlane((int)reinterpret_cast<const uint2 *>(scv.fwd_ptr)[row].y);
            if (pos == SG_PAIR_NO_CHUNK || (pos & 511u) + n_hit > SG_PAIR_CHUNK) {
                uint32_t c = 0;
                if (lane == 0) {
                    if (pos != SG_PAIR_NO_CHUNK && (pos >> 9) < pair_chunks) {
                        pv.d_chunk_count[pos >> 9] = pos & 511u;
                        atomicAdd(pv.d_totals, (unsigned long long)(pos & 511u));
                    }
                    c = atomicAdd(pv.d_chunks_used, 1u);
                }
                pos = (uint32_t)__builtin_amdgcn_readfirstlane((int)c) << 9;
            }
            if (((mm >> lane) & 1ull) && (pos >> 9) < pair_chunks) {   // past the capacity nothing is written: the caller falls back
                const size_t o = (size_t)(pos >> 9) * SG_PAIR_CHUNK + (pos & 511u) +
                                 ((uint32_t)__popcll(mm & ((1ull << lane) - 1ull)) << (both ? 1 : 0));
                pair_i[o] = row_name;
                pair_j[o] = (uint32_t)jo;
                pair_s[o] = sum;
                atomicAdd(&pair_row_count[jo], 1u);   // how many mirrored matches row j will receive (pass 2 scans these)
                if (both) {   // "row i receives column j"
                    pair_i[o + 1] = (uint32_t)jo;
                    pair_j[o + 1] = row_name;
                    pair_s[o + 1] = sum;
                }
            }
            if (both && lane == 0 && (pos >> 9) < pair_chunks) atomicAdd(&pair_row_count[row_name], n_hit >> 1);
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) surv[SG_SURV_CAP - 1] = (int)(pos + n_hit);
        }
        if (both) hm &= ~mm;   // (what is left: the diagonal, if the caller has not taken it)
    }
    SG_WD_DECL(wd_h);
    while (hm) {
        SG_WD(wd_h, 70, 22)
        const int src = __builtin_ctzll(hm);
        hm &= hm - 1;
        // (stream form: a column may be scored twice -- the filter's records are deduplicated by a small table that
        //  forgets -- and the same (score, column) must not enter the list twice)
        if (UNIQ) top.insert_unique(wave_read<T>(sum, src), wave_read<int>(jo, src), lane);
        else top.insert(wave_read<T>(sum, src), wave_read<int>(jo, src), lane);
    }
    if (n_surv > 64) {   // < 64 left: move to the front
        const uint32_t rem = n_surv - 64;
        const int keepv = (uint32_t)lane < rem ? surv[64 + lane] : 0;
        __builtin_amdgcn_wave_barrier();
        if ((uint32_t)lane < rem) surv[lane] = keepv;
    }
    return top;
}

// Stream form: the survivor buffer runs full.  Drops the columns recorded before -- a 128-entry table of the columns
// recorded last (ds_wrxchg); what it forgets is scored twice and dropped where results are kept -- closes the gaps and,
// if a full wave of survivors is left, scores it.  The first n_clean entries have been through this before (they ARE the
// table's entries) and stay.  Returns the list and, in .c of the extra lane-uniform word, the entries left (all clean).
// Not inlined: sixteen call sites in the round loop.  The caller has waited for its rounds in flight (this function and
// its callee save and restore registers).
template <typename T>
struct FlushOut {
    TopList<T> top;
    uint32_t n_surv;
};
template <typename T, bool SYM, int TILE_LOG2, bool WIDE>
__device__ __noinline__ FlushOut<T> flush_survivors(int nnz, T thr, uint32_t row, const SgScoreCtx *__restrict__ sc,
                                                    const SgPairSink *__restrict__ pairs, TopList<T> top, uint32_t n_surv,
                                                    uint32_t n_clean) {
    constexpr int TILE = 1 << TILE_LOG2;
    int *surv = lds_object<int>(TILE * 2 + 512 + 1024);
    uint32_t *dt = lds_object<uint32_t>(TILE * 2 + 512 + (sizeof(T) == 4 ? 512 : 1024 + SG_SURV_CAP * 4));
    const int lane = threadIdx.x;
    const uint64_t lanes_below = (1ull << lane) - 1ull;
    int c0 = 0, c1 = 0;
    bool k0 = false, k1 = false;
    if ((uint32_t)lane < n_surv) {
        c0 = surv[lane];
        k0 = (uint32_t)lane < n_clean ||
             __hip_atomic_exchange(&dt[((uint32_t)c0 * 2654435761u) >> 25], (uint32_t)c0, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_WORKGROUP) != (uint32_t)c0;
    }
    if ((uint32_t)lane + 64u < n_surv) {
        c1 = surv[lane + 64];
        k1 = (uint32_t)lane + 64u < n_clean ||
             __hip_atomic_exchange(&dt[((uint32_t)c1 * 2654435761u) >> 25], (uint32_t)c1, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_WORKGROUP) != (uint32_t)c1;
    }
    const uint64_t f0 = ballot64(k0), f1 = ballot64(k1);
    const uint32_t n0 = (uint32_t)__popcll(f0);
    __builtin_amdgcn_wave_barrier();
    if (k0) surv[__popcll(f0 & lanes_below)] = c0;
    if (k1) surv[n0 + __popcll(f1 & lanes_below)] = c1;
    __builtin_amdgcn_wave_barrier();
    FlushOut<T> out;
    out.top = top;
    out.n_surv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(n0 + (uint32_t)__popcll(f1)));
    // (from 63 on, not 64: the caller appends up to 64 columns behind what is left, and the buffer's last two words are not
    //  columns -- the pair list's position and the count of pairs scored exactly: 62 + 64 = 126)
    if (out.n_surv >= 63u) {
        out.top = drain_survivors<T, SYM, TILE_LOG2, WIDE, true>(nnz, thr, row, sc, pairs, top, out.n_surv);
        const uint32_t scored = out.n_surv < 64u ? out.n_surv : 64u;
        out.n_surv = (out.n_surv - scored) | (scored << 16);   // [0, 16) left in the buffer, [16, 32) handed to the scoring
    }
    return out;
}

// Launch over parts: the matches a part of a row kept (lane l: the l-th best, cnt of them) appended to the pair list as
// {column, row, score} -- "row receives column" -- through the wave's chunk like the mirrored pairs of drain_survivors.
template <typename T, int TILE_LOG2>
__device__ __noinline__ void emit_part_matches(const SgPairSink *__restrict__ pairs, T s, int c, int cnt, uint32_t row_out) {
    constexpr int TILE = 1 << TILE_LOG2;
    int *surv = lds_object<int>(TILE * 2 + 512 + 1024);
    const int lane = threadIdx.x;
    if (cnt <= 0) return;
    uint32_t pos = (uint32_t)surv[SG_SURV_CAP - 1];
    const SgPairSink pv = load_launch_constants(pairs);
    const uint32_t pair_chunks = pv.chunks;
    if (pos == SG_PAIR_NO_CHUNK || (pos & 511u) + (uint32_t)cnt > SG_PAIR_CHUNK) {
        uint32_t ch = 0;
        if (lane == 0) {
            if (pos != SG_PAIR_NO_CHUNK && (pos >> 9) < pair_chunks) {
                pv.d_chunk_count[pos >> 9] = pos & 511u;
                atomicAdd(pv.d_totals, (unsigned long long)(pos & 511u));
            }
            ch = atomicAdd(pv.d_chunks_used, 1u);
        }
        pos = (uint32_t)__builtin_amdgcn_readfirstlane((int)ch) << 9;
    }
    if ((pos >> 9) < pair_chunks) {   // past the capacity nothing is written: the caller falls back
        if (lane < cnt) {
            const size_t o = (size_t)(pos >> 9) * SG_PAIR_CHUNK + (pos & 511u) + (uint32_t)lane;
            pv.d_i[o] = (uint32_t)c;
            pv.d_j[o] = row_out;
            reinterpret_cast<T *>(pv.d_s)[o] = s;
        }
        if (lane == 0) atomicAdd(&pv.d_row_count[row_out], (uint32_t)cnt);
    }
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) surv[SG_SURV_CAP - 1] = (int)(pos + (uint32_t)cnt);
}

// WIDE: the second launch, over the rows the first one could not take because they have 65 .. 128 non-zeros: every lane
// stages two of the row's terms; still one posting list per lane, so the row's prefix P must fit 64 lanes.
// 16 single-wave workgroups per CU (the LDS limit) = 4 waves per SIMD: <= 128 VGPRs.  The f64 stream form has 10.5 KiB of LDS
// per wave (15 per CU) and, at 128 registers, spills the visit ends inside its round loop -- scratch reloads that the
// compiler waits for with vmcnt(0), i.e. the rounds in flight drained every round: it is built for 3 waves per SIMD.
// SHARE: a launch over a rank's share of the rows, or over the parts of the rows such a launch has set aside (stream +
// self-join form only).  The whole-matrix pass is an instantiation of its own: what the parts need alive across the round
// loop -- the list of rows set aside, the end of the visits as a variable -- costs it scalar registers it does not have
// (49 instead of 20 spilled, + 0.8 % at 663 k).
template <typename T, int TILE_LOG2, bool SYM, bool WIDE, int FOLD_LOG2, bool SHARE = false>
__global__ void __launch_bounds__(64, (sizeof(T) == 8 && FOLD_LOG2 > 0) ? SG_F64_STREAM_WAVES : 4)
spgemm_topn_pruned_kernel(const int64_t *__restrict__ a_indptr, const int32_t *__restrict__ a_indices,
                          const T *__restrict__ a_data, uint32_t n_left, const uint32_t *__restrict__ seg,
                          const uint32_t *__restrict__ ends, int32_t nt_pad, uint32_t n_terms,
                          const uint32_t *__restrict__ filt, int32_t n_tiles,
                          const SgScoreCtx *__restrict__ sc /* packed rows of B and the position -> row table (sg_internal.h) */,
                          int32_t keep, int32_t out_stride, T thr,
                          float s_budget /* (beta / max ||b_j||)^2, rounded down */, float norm_b /* max ||b_j||, rounded up */,
                          uint32_t freq_min /* list length from which a term may join the suffix */,
                          int32_t *__restrict__ out_cols, T *__restrict__ out_vals, int32_t *__restrict__ out_cnt,
                          uint32_t *row_counter, uint32_t *flagged_count, uint32_t *flagged_rows,
                          unsigned long long *stats /* [0] rows [1] postings streamed [2] survivors [4] pairs scored exactly */,
                          const SgPairSink *__restrict__ pairs /* SYM: the pair list (sg_internal.h), in device memory */,
                          uint32_t pair_chunks /* chunks there are */,
                          uint32_t sym_lo, uint32_t sym_hi /* SYM: the left rows this launch scores: sym_hi - 1, sym_hi - 1 - sym_step, ...
                                                              >= sym_lo (multi-GPU: a rank's share; sym_step = 1: a contiguous range) */,
                          const uint32_t *__restrict__ row_list /* WIDE: the rows to process */, const uint32_t *row_list_len,
                          const uint32_t *__restrict__ ends8 /* stream form: ends of the super-tiles */, int32_t nv_pad,
                          uint32_t null_off /* stream form: byte offset of 256 filter postings that add nothing, four per lane */,
                          uint32_t n_right /* right-hand rows (columns of the result) */,
                          uint32_t *heavy_count, uint32_t *heavy_rows /* stream + self-join form: rows set aside for the launch over parts */,
                          uint32_t sym_step,
                          uint32_t part_cfg /* 0: every row whole; < 2^31: rows of at least this many rounds (bits [0, 28)) are set aside, and a row that has scored rounds << bits [28, 31) candidates hands its remaining visits on;
                                               bit 31: this IS the launch over parts (rows in row_list, SG_ROW_PARTS items each) */) {
    constexpr int TILE = 1 << TILE_LOG2;
    constexpr int SLOTS = WIDE ? 2 : 1;   // row terms staged per lane
    constexpr int AB = TILE_LOG2 + 1;                          // address + half bits of a filter posting
    constexpr uint32_t ADDR_MASK = ((1u << AB) - 1u) & ~3u;    // byte address of the accumulator word
    // stream form (FOLD_LOG2 > 0): 2^FOLD_LOG2 tiles share the accumulator tile; the posting's fold field sits between
    // the address and bq, and the fixed point is coarser by the same factor so that an accumulator holding the sums of
    // all the columns folded onto it still fits sixteen bits
    constexpr int FB = AB + FOLD_LOG2;                         // first bit of bq
    constexpr uint32_t BQ_BITS = 24 - FB;
    constexpr uint32_t BQ_MAX = (1u << BQ_BITS) - 1u;
    constexpr float SCALE = (float)(32768 >> FOLD_LOG2);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // smem: TILE u16 accumulators (address 0), then
    int *hk = reinterpret_cast<int *>(smem + TILE * 2);                       // row i: hash of its terms
    T *ha = reinterpret_cast<T *>(smem + TILE * 2 + 512);                     //        and their values
    int *surv = reinterpret_cast<int *>(smem + TILE * 2 + 512 + 1024);        // SG_SURV_CAP columns
    uint4 *tab_v = reinterpret_cast<uint4 *>(smem);
    const int lane = threadIdx.x;
    for (int x = lane; x < TILE * 2 / 16; x += 64) tab_v[x] = make_uint4(0, 0, 0, 0);
    if (SYM && lane == 0) surv[SG_SURV_CAP - 1] = (int)SG_PAIR_NO_CHUNK;
    if (FOLD_LOG2 > 0 && lane == 0) surv[SG_SURV_CAP - 2] = 0;   // stream form: pairs this wave scores exactly (drain_survivors)
    const uint64_t lanes_below = (1ull << lane) - 1ull;
    unsigned long long st_rows = 0, st_post = 0, st_surv = 0;
    // the accumulator tile is the kernel's first LDS object (address 0): a posting's address field IS the LDS address
    auto tab_at = [&](uint32_t byte_addr) {
        typedef __attribute__((address_space(3))) uint32_t lds_u32;
        return (uint32_t *)(lds_u32 *)(uintptr_t)byte_addr;
    };
    auto filt_at = [&](uint32_t byte_off) {   // kernel-constant base + 32-bit byte offset
        return *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(filt) + byte_off);
    };

    SG_WD_DECL(wd_rows);
    // Launch over PARTS (stream + self-join form, sg_spgemm_pruned_symmetric): an item is one sixteenth of the visits of a row
    // the first launch set aside -- a row of thousands of rounds is a chain of dependent loads in ONE wave (3.4 ms for the
    // slowest row of the 663 k job when the whole pass takes 6), which is what a rank's range of the multi-GPU form ends
    // with (scripts/range_probe.py).  A part keeps its matches in its list like a row and hands them to the pair list,
    // addressed to its own row; the second pass merges the parts' lists like mirrored matches.
    constexpr bool CAN_SPLIT = SYM && !WIDE && FOLD_LOG2 > 0 && SHARE;
    const bool part_mode = CAN_SPLIT && (part_cfg >> 31) != 0u;
#ifdef SG_DEBUG_WAVE_TIMES
    const unsigned long long dbg_t0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long dbg_worst = 0, dbg_worst_row = 0;
#endif
    const uint32_t n_here = (WIDE || part_mode) ? (uint32_t)__builtin_amdgcn_readfirstlane((int)*row_list_len) << (part_mode ? SG_ROW_PARTS_LOG2 : 0)
                                                : (SYM ? (sym_hi - sym_lo + sym_step - 1u) / sym_step : n_left);
    // rows are handed out four at a time: one global atomic per row capped the kernel at ~88 rows/us (larger first
    // helpings were tried -- sixteen rows for the first three quarters -- and changed nothing: profiles/r02_sessionJ6_*.log).
    // Symmetric mode walks the rows from the last to the first: a row's cost grows with its index there.
    // The last rows -- two per wave of the launch -- go one at a time: a launch ends with its slowest helping, and when
    // the launch is a rank's RANGE of the multi-GPU self-join the last rows are not the cheap ones of the whole job but
    // rows as expensive as the range's first (scripts/sim_scaling.py: the eight ranges of the 663 k job took 25.8 ms in
    // all against 10.0 ms for the whole -- tails of four-row helpings).
    const uint32_t n_single = min(n_here, 2u * gridDim.x);
    const uint32_t n_quads = (n_here - n_single) >> 2;     // helpings [0, n_quads) hold four rows, the rest one
    for (uint32_t hlp = next_row(row_counter, lane);; hlp = next_row(row_counter, lane)) {
    const uint32_t row0 = hlp < n_quads ? hlp << 2 : (n_quads << 2) + (hlp - n_quads);
    if (row0 >= n_here || hlp >= 0x10000000u) break;       // (the second: the pass was called off, see the pair list below)
    for (uint32_t rr = row0; rr < row0 + (hlp < n_quads ? 4u : 1u); ++rr) {   // (rows behind the quads go singly)
        SG_WD(wd_rows, n_left + 2, 11)
        const uint32_t row = WIDE ? (uint32_t)__builtin_amdgcn_readfirstlane((int)row_list[rr])
                                  : part_mode ? (uint32_t)__builtin_amdgcn_readfirstlane((int)row_list[rr >> SG_ROW_PARTS_LOG2])
                                              : (SYM ? sym_hi - 1u - rr * sym_step : rr);
        uint32_t part_lo = 0, part_hi = 0;   // part mode: the visits [part_lo, part_hi) of the row
        if (CAN_SPLIT && part_mode) {
            // (the visits in front of `from` the row has done itself, in the launch over rows: see `deferred` below)
            const uint32_t from = (uint32_t)__builtin_amdgcn_readfirstlane((int)row_list[n_left + (rr >> SG_ROW_PARTS_LOG2)]);
            const uint32_t nv = (((row >> TILE_LOG2) + (1u << FOLD_LOG2)) >> FOLD_LOG2) - from;   // visits of the row (self-join form) left
            const uint32_t part = rr & (SG_ROW_PARTS - 1u);
            part_lo = from + ((part * nv) >> SG_ROW_PARTS_LOG2);
            part_hi = from + (((part + 1u) * nv) >> SG_ROW_PARTS_LOG2);
            if (part_lo == part_hi) continue;
        }
        // self-join form: the left matrix IS the permuted one, `row` a position; its result row and its name in the pairs
        // are the original row's.  (One-sided form: the left rows are the caller's, only the columns are positions.)
        uint32_t row_out = row;
        if (SYM) {
            const uint32_t *orig_of = sc->orig_of;   // position -> row (sg_postings.hip); null: identity
            if (orig_of) row_out = (uint32_t)__builtin_amdgcn_readfirstlane((int)orig_of[row]);
        }
        const int64_t rlo = a_indptr[row];
        const int nnz = __builtin_amdgcn_readfirstlane((int)(a_indptr[row + 1] - rlo));
        if (nnz > 64 * SLOTS) {   // more non-zeros than this launch stages: the wide launch, or the exact kernel
            if (lane == 0) flagged_rows[atomicAdd(flagged_count, 1u)] = row;
            continue;
        }
        if (nnz == 0) continue;   // out_cnt is zero-initialised
        int k[SLOTS];
        T a[SLOTS];
        uint32_t df[SLOTS], lo_e[SLOTS];
        float w[SLOTS], cum[SLOTS];
#pragma unroll
        for (int sl = 0; sl < SLOTS; ++sl) {
            k[sl] = 0;
            a[sl] = (T)0;
            df[sl] = lo_e[sl] = 0;
            if (lane + 64 * sl < nnz) {
                k[sl] = a_indices[rlo + lane + 64 * sl];
                a[sl] = a_data[rlo + lane + 64 * sl];
                const uint32_t *sp = seg + (int64_t)k[sl] * n_tiles;
                lo_e[sl] = sp[0];
                df[sl] = sp[n_tiles] - lo_e[sl];
            }
            // rounded up: covers the float sums below
            w[sl] = lane + 64 * sl < nnz ? (float)a[sl] * (float)a[sl] * 1.00001f : 0.f;
            cum[sl] = 0.f;
        }
        // ---- suffix S: the most frequent terms while the bound on ||a_S|| holds.  cum = sum of squares of
        // the terms ordered before this one (list length descending, position ascending), inclusive.
        for (int q = 0; q < nnz; ++q) {
            uint32_t dq;
            float wq;
            if (SLOTS == 1 || q < 64) {
                dq = wave_read<uint32_t>(df[0], q);
                wq = wave_read<float>(w[0], q);
            } else {
                dq = wave_read<uint32_t>(df[SLOTS - 1], q - 64);
                wq = wave_read<float>(w[SLOTS - 1], q - 64);
            }
#pragma unroll
            for (int sl = 0; sl < SLOTS; ++sl) {
                const bool before = dq > df[sl] || (dq == df[sl] && q <= lane + 64 * sl);
                cum[sl] += before ? wq : 0.f;
            }
        }
        bool in_p[SLOTS];
        uint64_t pm[SLOTS];
        float bs2 = 0.f, dsum = 0.f;
        int np = 0;
#pragma unroll
        for (int sl = 0; sl < SLOTS; ++sl) {
            const bool have = lane + 64 * sl < nnz;
            const bool in_s = have && cum[sl] <= s_budget && df[sl] >= freq_min;   // a prefix of the (df desc) order
            in_p[sl] = have && !in_s;
            pm[sl] = __ballot(in_p[sl]);
            np += __popcll(pm[sl]);
            bs2 = fmaxf(bs2, in_s ? cum[sl] : 0.f);
            dsum += in_p[sl] ? (float)df[sl] : 0.f;
        }
        if (np == 0) continue;   // ||a|| * max ||b|| <= beta < threshold: no match possible
        if (np > 64) {           // (wide rows only) more prefix terms than lanes: exact kernel
            if (lane == 0) flagged_rows[atomicAdd(flagged_count, 1u)] = row;
            continue;
        }
        if (!part_mode) ++st_rows;
#ifdef SG_DEBUG_WAVE_TIMES
        const unsigned long long dbg_r0 = __builtin_amdgcn_s_memrealtime();
        const unsigned long long dbg_surv0 = st_surv;
#endif
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            bs2 = fmaxf(bs2, __shfl_xor(bs2, d, 64));
            dsum += __shfl_xor(dsum, d, 64);
        }
        bs2 = wave_read<float>(bs2, 0);     // explicitly wave-uniform: the branches below must not diverge
        dsum = wave_read<float>(dsum, 0);
        if (!(dsum > 0.f)) continue;   // every list of P is empty
        // ---- survivor test in fixed point (scale 2^15).  q_ij accumulates UPPER bounds of the products, so
        //      p_ij * 2^15 <= q_ij; the exact kernel's float score obeys  score~ <= score + 1e-5  and
        //      score <= p_ij + ||a_S|| f_j  with  f_j <= fq_j / 255 * norm_b.  Column j survives when
        //      q_ij >= tq_j,  tq_j = floor(((T0 - C1 * fq_j)) / 256)  <=  (thr - 1e-5) * 2^15 - 2 - c1 * fq_j
        //      (T0 = floor(t0 * 256), C1 = floor(c1 * 256) + 1; the 2 covers the float evaluation of t0, c1).
        const float b_s = sqrtf(bs2) * 1.000002f;
        const float t0 = ((float)thr - 1e-5f) * SCALE - 2.0f;
        const float c1 = b_s * norm_b * (SCALE / 255.0f) * 1.000002f;
        const int32_t T0 = (int32_t)floorf(t0 * 256.0f) - 256 * np;   // every add may fall short by < 1, |P| adds at most
        const int32_t C1 = (int32_t)(c1 * 256.0f) + 1;
        // Stream form: the posting's bits [16, 32) AS ONE NUMBER F (fq with bq as its low byte, stored as F - 32768: K3 rounds
        // fq so that F is an upper bound of f_j / norm_up * 255 * 256) give the threshold in sixteen more bits,
        //      d_j = T16 - C16 * F  <=  tq_j * 2^16,      survive when (q_ij << 16) >= d_j,
        // in ONE instruction: v_mad_i32_i16 on the posting's upper half, d_j = T0s + C1n * int16(F - 32768).
        const int32_t C16 = (int32_t)(b_s * norm_b * (SCALE * 65536.0f / 65280.0f) * 1.000002f) + 1;   // < 2^13
        const int32_t T16 = T0 << 8;                                                                     // < 2^29
        const int32_t T0s = T16 - 32768 * C16, C1n = -C16;
        if (FOLD_LOG2 > 0 ? !(T16 - C16 * 65535 >= 65536) : !(T0 - C1 * 255 >= 256)) {   // delta too small for the fixed-point resolution: exact kernel
            if (lane == 0) flagged_rows[atomicAdd(flagged_count, 1u)] = row;
            continue;
        }
        const int32_t T0m = T0 - 256;    // (T0m - C1 * fq) >> 8 == tq - 1 >= 0

        // ---- deal the 64 lanes to the terms of P in proportion to their list lengths
        uint32_t G[SLOTS], start[SLOTS];
        uint32_t before_slot = 0;
#pragma unroll
        for (int sl = 0; sl < SLOTS; ++sl) {
            G[sl] = in_p[sl] ? 1u + (uint32_t)((float)(64 - np) * 0.999f * ((float)df[sl] / dsum)) : 0u;
            uint32_t inc = G[sl];   // inclusive scan, made exclusive below
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t o = __shfl_up(inc, d, 64);
                if (lane >= d) inc += o;
            }
            start[sl] = before_slot + inc - G[sl];
            before_slot += wave_read<uint32_t>(inc, 63);
        }
        int src = 0, src_slot = 0;
        uint32_t u = 0, g = 0;
#pragma unroll
        for (int sl = 0; sl < SLOTS; ++sl) {
            uint64_t m = pm[sl];
            SG_WD_DECL(wd_a);
            while (m) {
                SG_WD(wd_a, 70, 12)
                const int f = __builtin_ctzll(m);
                m &= m - 1;
                const uint32_t sf = wave_read<uint32_t>(start[sl], f), gf = wave_read<uint32_t>(G[sl], f);
                const uint32_t d = (uint32_t)lane - sf;
                if (d < gf) {
                    src = f;
                    src_slot = sl;
                    u = d;
                    g = gf;
                }
            }
        }
        int my_k = wave_shfl<int>(k[0], src);
        uint32_t my_lo = wave_shfl<uint32_t>(lo_e[0], src);
        T my_a = wave_shfl<T>(a[0], src);
        if (SLOTS > 1) {
            const int k1 = wave_shfl<int>(k[SLOTS - 1], src);
            const uint32_t lo1 = wave_shfl<uint32_t>(lo_e[SLOTS - 1], src);
            const T a1 = wave_shfl<T>(a[SLOTS - 1], src);
            if (src_slot) {
                my_k = k1;
                my_lo = lo1;
                my_a = a1;
            }
        }
        // upper bound (less one) of a * b * 2^15 from the bq field of a filter posting, left in place:
        // x = (CA * (bq << AB)) >> 32 with CA >= c_a * 2^(32 - AB), c_a = a * norm_b * 2^15 / BQ_MAX
        const float c_a = (float)my_a * norm_b * (SCALE / (float)BQ_MAX) * 1.000002f;
        const uint32_t CA = (uint32_t)(c_a * (float)(1u << (32 - FB))) + 1u;   // < 2^24
        // lane (term, u of G): byte offset of its first entry inside a segment, stride; idle lanes read the
        // all-zero row K3 appends to the table of segment ends (empty segments, rem == 0) and entry 0
        // Lane u of the g lanes of a term owns entries 4u .. 4u + 3 of the term's segment in every tile: one 16-byte load.
        // (Four dword loads at stride g were the first layout: the same bytes, four times the work for the address unit,
        // which was the busiest block of the kernel -- profiles/r02_sessionB_*.log, TCP_GATE_EN ~ 100 %.)  What a segment
        // holds beyond 4g entries goes slot by slot (round), lane u taking entries 4g + u, 4g + u + g, ...
        const uint32_t u4 = u << 2, u16 = u << 4, G4 = g << 2, G16 = g << 4;
        const int32_t over16 = (int32_t)(G16 - u16);   // rem > over16: the segment holds more than 4g entries
        const uint32_t erow = (g ? (uint32_t)my_k : n_terms) * (uint32_t)nt_pad;
        auto ends_at = [&](uint32_t group) {
            return *reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(ends) + ((erow + (group << 2)) << 2));
        };
        const uint32_t t_end = SYM ? (row >> TILE_LOG2) + 1u : (uint32_t)n_tiles;
        const uint32_t last_group = (uint32_t)(nt_pad >> 2) - 1u;

        // ---- stage row i for the exact scoring
        if (WIDE) {   // the sorted row itself: terms in hk, values in ha (binary search, row_value)
#pragma unroll
            for (int sl = 0; sl < SLOTS; ++sl) {
                hk[lane + 64 * sl] = lane + 64 * sl < nnz ? k[sl] : INT32_MAX;
                ha[lane + 64 * sl] = a[sl];
            }
            __builtin_amdgcn_wave_barrier();
        } else {      // term -> value hash (filled by compare-and-swap, one wave)
            hk[lane] = -1;
            hk[lane + 64] = -1;
            __builtin_amdgcn_wave_barrier();
            if (lane < nnz) {
                uint32_t h = term_hash(k[0]);
                SG_WD_DECL(wd_i);
                while (atomicCAS(&hk[h], -1, k[0]) != -1) {
                    SG_WD(wd_i, SG_HASH_SLOTS + 2, 16)
                    h = (h + 1) & (SG_HASH_SLOTS - 1);
                }
                ha[h] = a[0];
            }
            __builtin_amdgcn_wave_barrier();
        }

        TopList<T> top;
        top.clear();
        uint32_t n_surv = 0;
        // Self-join, stream form: the row's match with ITSELF comes from the registers -- the sum of its squares in ascending
        // term order, product and sum rounded separately: what the exact scoring computes for the pair (i, i), bit for bit
        // (row i of B holds the same values, and entries a round reads past a row's end add a * 0).  With the second filter
        // in front of the exact scoring the diagonal was the one candidate of most rows that passed it: a second chain of
        // dependent misses (packed row, its rounds) at the end of nearly every row, for a number the wave already holds.
        // The survivor routine is told to leave the diagonal alone (bit 31 of its row argument).  Rows worked off in parts
        // keep the old way (the part that holds the row's own tile scores it); a row that hands its last visits to the
        // parts has the pair twice in the pair list's merge, which drops repeats (pairs_select_kernel).
        uint32_t row_arg = row | ((SYM && keep > SG_TOPN_LANES) ? 0x40000000u : 0u);   // (bit 30: see drain_survivors)
        if (SYM && FOLD_LOG2 > 0 && !part_mode) {
            T own = (T)0;
            for (int q = 0; q < nnz; ++q) {
                const T aq = (SLOTS == 1 || q < 64) ? wave_read<T>(a[0], q) : wave_read<T>(a[SLOTS - 1], q - 64);
                own = add_rn<T>(own, mul_rn<T>(aq, aq));
            }
            if (own > thr) top.insert(own, (int)row_out, lane);
            row_arg |= 0x80000000u;
        }

        // The survivors of one slot of a tile (rare: a few per row): append the crossing lanes' columns, score a
        // full wave of them at once.
        auto collect = [&](uint64_t cm, uint32_t r, uint32_t t) {
            bool cross = (cm >> lane) & 1ull;
            const int col = (int)((t << TILE_LOG2) | ((r & ADDR_MASK) >> 1) | (r & 1u));
            if (SYM) {
                cross = cross && (uint32_t)col <= row;   // the pair (i, j > i) is row j's to score
                cm = ballot64(cross);
            }
            if (cross) surv[n_surv + __popcll(cm & lanes_below)] = col;
            n_surv += __popcll(cm);
            if (n_surv >= 64) {
                top = drain_survivors<T, SYM, TILE_LOG2, WIDE, (FOLD_LOG2 > 0)>(nnz, thr, row_arg, sc, pairs, top, n_surv);
                st_surv += 64;
                n_surv -= 64;
            }
        };
        // One posting applied by the lanes in `valid`, waiting for its result (the slot-by-slot form, used for tiles
        // with more than four postings per lane): the add into the 16-bit accumulator and the survivor test.
        // (Only the atomic is under the lane mask: everything else is side-effect free and runs for all lanes, which
        // keeps the survivor mask a wave-uniform scalar for the compiler.)
        auto round = [&](uint32_t r, bool valid, uint64_t valid_mask, uint32_t t, uint32_t &zaddr) {
            zaddr = r & ADDR_MASK;
            const uint32_t sh = r << 4;   // bit 4 = the half; shifts and bit-field offsets use 5 bits
            const uint32_t x = (uint32_t)(((uint64_t)(r & (BQ_MAX << FB)) * (uint64_t)(CA & 0xffffffu)) >> 32);
            const uint32_t tq1 = (uint32_t)((T0m - __mul24(C1, (int32_t)(r >> 24))) >> 8);
            uint32_t xs;   // x << (16 * half): the hardware shift takes the low five bits of sh by itself
            asm("v_lshlrev_b32 %0, %1, %2" : "=v"(xs) : "v"(sh), "v"(x));
            uint32_t old;
            asm("" : "=v"(old));   // lanes without a posting: whatever the register holds, masked below
            if (valid)
                old = __hip_atomic_fetch_add(tab_at(zaddr), xs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            asm volatile("" : "+v"(old));   // keep the test outside the masked region: its result is then a plain lane mask
            const uint32_t oh = __builtin_amdgcn_ubfe(old, sh, 16u);
            // oh < tq && oh + x >= tq  (unsigned wrap when oh >= tq); the mask of a compare ANDed with a scalar mask
            // stays scalar (the mask of a combined predicate would be rebuilt through a VALU select)
            const uint64_t cm = ballot64(tq1 - oh < x) & valid_mask;
            if (cm) collect(cm, r, t);
        };

        // a batch = the lane's (up to) four entries of one tile's segment + what is left of the segment from the
        // lane's first entry on, in bytes (<= 0: nothing for this lane)
        struct Batch {
            uint32_t r0, r1, r2, r3;
            int32_t rem;
            uint32_t base;
        };
        auto issue = [&](Batch &bt, uint32_t lo, uint32_t hi) {
            bt.base = lo + u16;
            bt.rem = (int32_t)(hi - bt.base);
            // unconditional on purpose: loading only for the lanes whose slots hold postings (60 % are empty) would put the
            // loads under branches, and the compiler then loses its exact count of loads in flight -- it waits for (almost)
            // all of them before a batch is used, i.e. the prefetch distance is gone: +9 % (profiles/r02_sessionI_*.log)
            struct __attribute__((packed, aligned(4))) Quad {
                uint32_t x, y, z, w;
            };
            // (pointing the lanes without a posting at one common line instead made the kernel 2.5 x slower:
            // profiles/r02_sessionR_idle_lanes_one_line.log; loads under an exec mask written in inline assembly, with
            // hand-counted waits, changed nothing: profiles/r02_sessionT_masked_asm_loads.log)
            const Quad q = *reinterpret_cast<const Quad *>(reinterpret_cast<const char *>(filt) + bt.base);
            bt.r0 = q.x;
            bt.r1 = q.y;
            bt.r2 = q.z;
            bt.r3 = q.w;
        };
        // What one slot adds and what its accumulator must reach: side-effect free, computed for all lanes.
        struct Slot {
            uint32_t z, sh, x, xs, tq1;
        };
        auto prep = [&](uint32_t r) {
            Slot s;
            s.z = r & ADDR_MASK;
            s.sh = r << 4;   // bit 4 = the half; shifts and bit-field offsets use 5 bits
            s.x = (uint32_t)(((uint64_t)(r & (BQ_MAX << FB)) * (uint64_t)(CA & 0xffffffu)) >> 32);
            s.tq1 = (uint32_t)((T0m - __mul24(C1, (int32_t)(r >> 24))) >> 8);
            // x << (16 * half): the hardware shift takes the low five bits of sh by itself
            asm("v_lshlrev_b32 %0, %1, %2" : "=v"(s.xs) : "v"(s.sh), "v"(s.x));
            return s;
        };
        // One tile: the (up to) four postings of every lane.  All four adds and the four re-zeroing stores are
        // issued back to back -- DS operations of a wave execute in order, so an add's returned value reflects
        // every earlier add and the stores land after all of them -- and the wave waits ONCE for the returns
        // (round 2's first version waited per slot: 2.5 LDS round trips per tile made the loop latency-bound,
        // profiles/r02_sessionA_*.log).  Everything but the LDS operations runs for all lanes.
        auto apply = [&](const Batch &bt, uint32_t t) {
            const bool v0 = bt.rem > 0, v1 = bt.rem > 4, v2 = bt.rem > 8, v3 = bt.rem > 12;
            const uint64_t m0 = ballot64(v0), m1 = ballot64(v1), m2 = ballot64(v2), m3 = ballot64(v3);   // (next to the compares:
            if (m0 == 0) return;                                                  //  taken across a branch, a mask is rebuilt through a VGPR)
            if (ballot64(bt.rem > over16)) {
                // a segment longer than four entries per lane (rare): slot by slot, then a full clear
                uint32_t zz = 0;
                round(bt.r0, v0, m0, t, zz);
                round(bt.r1, v1, m1, t, zz);
                round(bt.r2, v2, m2, t, zz);
                round(bt.r3, v3, m3, t, zz);
                int32_t left = bt.rem - over16 - (int32_t)u4;
                uint32_t at = bt.base + (uint32_t)over16 + u4;
                SG_WD_DECL(wd_b);
                uint64_t ml;
                while ((ml = ballot64(left > 0)) != 0) {
                    SG_WD(wd_b, 1 << 24, 14)
                    const uint32_t rx = filt_at(at);
                    round(rx, left > 0, ml, t, zz);
                    left -= (int32_t)G4;
                    at += G4;
                }
                for (int x = lane; x < TILE * 2 / 16; x += 64) tab_v[x] = make_uint4(0, 0, 0, 0);
                return;
            }
            const Slot s0 = prep(bt.r0), s1 = prep(bt.r1), s2 = prep(bt.r2), s3 = prep(bt.r3);
            // No lane mask on the eight LDS operations: a slot without a posting holds whatever lies behind the segment, i.e.
            // SOME accumulator address of this tile; it adds 0 there, and its re-zeroing store is harmless because this batch
            // is the row's whole visit of the tile -- every accumulator it touches ends at zero anyway and DS operations
            // execute in order, after all four adds.  (Under masks every operation cost a saveexec, two taken branches and a
            // restore: 16 branches per tile.)
            const uint32_t a0 = v0 ? s0.xs : 0u, a1 = v1 ? s1.xs : 0u, a2 = v2 ? s2.xs : 0u, a3 = v3 ? s3.xs : 0u;
            uint32_t o0 = __hip_atomic_fetch_add(tab_at(s0.z), a0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            uint32_t o1 = __hip_atomic_fetch_add(tab_at(s1.z), a1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            uint32_t o2 = __hip_atomic_fetch_add(tab_at(s2.z), a2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            uint32_t o3 = __hip_atomic_fetch_add(tab_at(s3.z), a3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            // re-zero only what was touched (a few dozen of the tile's accumulators): sweeping the tile for every
            // row costs rows * columns * 2 B of LDS writes, 11 ms at 663 k
            *tab_at(s0.z) = 0u;
            *tab_at(s1.z) = 0u;
            *tab_at(s2.z) = 0u;
            *tab_at(s3.z) = 0u;
            asm volatile("" : "+v"(o0), "+v"(o1), "+v"(o2), "+v"(o3));   // the one wait; the tests below stay outside the masks
            // oh < tq && oh + x >= tq  (unsigned wrap when oh >= tq); the mask of a compare ANDed with a scalar mask
            // stays scalar (the mask of a combined predicate would be rebuilt through a VALU select)
            const uint64_t c0 = ballot64(s0.tq1 - __builtin_amdgcn_ubfe(o0, s0.sh, 16u) < s0.x) & m0;
            const uint64_t c1m = ballot64(s1.tq1 - __builtin_amdgcn_ubfe(o1, s1.sh, 16u) < s1.x) & m1;
            const uint64_t c2 = ballot64(s2.tq1 - __builtin_amdgcn_ubfe(o2, s2.sh, 16u) < s2.x) & m2;
            const uint64_t c3 = ballot64(s3.tq1 - __builtin_amdgcn_ubfe(o3, s3.sh, 16u) < s3.x) & m3;
            if (c0 | c1m | c2 | c3) {
                if (c0) collect(c0, bt.r0, t);
                if (c1m) collect(c1m, bt.r1, t);
                if (c2) collect(c2, bt.r2, t);
                if (c3) collect(c3, bt.r3, t);
            }
        };

        if constexpr (FOLD_LOG2 == 0) {
            // Software pipeline, four tiles per trip: E = segment ends of tiles 4m .. 4m + 3 (byte offsets),
            // prev = end of tile 4m - 1 (= start of tile 4m); batch j of a trip belongs to tile 4m + j and is
            // re-issued for tile 4m + j + 4 ... no: the batch of tile t + 3 is issued while tile t is applied.
            uint4 E0 = ends_at(0);
            uint4 E1 = ends_at(min(1u, last_group));
            const uint32_t list_lo = g ? my_lo << 2 : 0u;
            Batch b0, b1, b2, b3;
            issue(b0, list_lo, E0.x);
            issue(b1, E0.x, E0.y);
            issue(b2, E0.y, E0.z);
            SG_WD_DECL(wd_t);
            for (uint32_t t = 0; t < t_end; t += 4) {
                SG_WD(wd_t, n_tiles + 2, 13)
                // tiles t .. t + 3 use E0; E1 = the next four; E2 is fetched for the trip after
                const uint4 E2 = ends_at(min((t >> 2) + 2u, last_group));
                issue(b3, E0.z, E0.w);
                apply(b0, t);
                if (t + 1 >= t_end) break;
                issue(b0, E0.w, E1.x);
                apply(b1, t + 1);
                if (t + 2 >= t_end) break;
                issue(b1, E1.x, E1.y);
                apply(b2, t + 2);
                if (t + 3 >= t_end) break;
                issue(b2, E1.y, E1.z);
                apply(b3, t + 3);
                E0 = E1;
                E1 = E2;
            }
        } else {
            // ---- Stream form (round 3).  The tile-by-tile loop above pays ~150 instructions per (row, tile) visit for ~110
            // postings in 256 slots: lanes are dealt to the terms once per row, a tile's segment of a term is short, and
            // what a lane's four slots do not hold goes through a slow path (scripts/k4f_fold_model.py: 34 % of the slots
            // used, 11 % of the visits overflow).  Here a visit covers a SUPER-TILE of 2^FOLD_LOG2 tiles whose columns share
            // one accumulator tile (column c -> accumulator c mod TILE; K3 writes the tile index mod 2^FOLD_LOG2 into the
            // posting) and is worked off in ROUNDS of one 16-byte load per lane -- as many as the longest lane stream of the
            // visit needs -- with the loads of the next three rounds in flight, whichever visit they belong to.  Eight
            // times fewer visits, twice the postings per round (the relative spread of a segment's length shrinks with its
            // size), no slow path.
            //
            // Sums of several columns in one accumulator are still UPPER bounds of each of them, so nothing is lost as long
            // as every posting that finds its accumulator at or above ITS column's survivor threshold records the column:
            // the test is `old + x >= tq_j`, not the crossing `old < tq_j <= old + x` (with two columns in an accumulator
            // the crossing may be another column's doing).  Price: false positives (two unrelated partial sums adding up;
            // + 20 % at 663 k) and repeats (every posting of a column after the first that passes); both only cost an exact
            // scoring.  Repeats are caught by a 128-entry table of the columns recorded last (ds_wrxchg); what it forgets
            // is scored twice and dropped where results are kept (TopList::insert_unique, pairs_select_kernel).
            // Accumulators are cleared once per visit (8 KiB, eight ds_write_b128 per lane) instead of per posting: with
            // more than one round the re-zeroing stores of a round would wipe sums that later rounds add to.
            constexpr int FOLD = 1 << FOLD_LOG2;
            constexpr uint32_t COL_MASK = (1u << (TILE_LOG2 + FOLD_LOG2)) - 1u;
            uint32_t *dt = reinterpret_cast<uint32_t *>(smem + TILE * 2 + 512 + (sizeof(T) == 4 ? 512 : 1024 + SG_SURV_CAP * 4));
            dt[lane] = 0xFFFFFFFFu;
            dt[lane + 64] = 0xFFFFFFFFu;
            const uint32_t n_visits = (t_end + (uint32_t)FOLD - 1u) >> FOLD_LOG2;
            const uint32_t erow8 = (g ? (uint32_t)my_k : n_terms) * (uint32_t)nv_pad;
            auto ends8_at = [&](uint32_t group) {
                return *reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(ends8) + ((erow8 + (group << 2)) << 2));
            };
            const uint32_t last_group8 = (uint32_t)(nv_pad >> 2) - 1u;
            // the stream of a lane ends with the row's last tile (self-join form: the row's own tile), which need not be the
            // end of a super-tile
            const uint32_t hi_end = ends[(size_t)erow + t_end - 1u];
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            uint32_t v_lo = 0, v_end = n_visits;              // the visits this item covers
            uint32_t hi_stop = hi_end;                        // where the lane's stream ends
            uint32_t cur = (g ? my_lo << 2 : 0u) + u16;       // the lane's next four entries
            if (CAN_SPLIT) {
                if (part_mode) {
                    v_lo = part_lo;
                    v_end = part_hi;
                    const uint32_t *e8 = ends8 + erow8;
                    if (v_end < n_visits) hi_stop = min(hi_end, e8[v_end - 1u]);
                    if (v_lo) cur = e8[v_lo - 1u] + u16;
                } else if (part_cfg != 0u && n_visits >= 2u) {
                    // rounds of the row = the longest lane stream; from part_cfg rounds on the row is set aside
                    float rl = g ? (float)(hi_end - (my_lo << 2)) / (float)G16 : 0.f;
#pragma unroll
                    for (int d = 32; d > 0; d >>= 1) rl = fmaxf(rl, __shfl_xor(rl, d, 64));
                    if (wave_read<float>(rl, 0) >= (float)(part_cfg & 0x0fffffffu)) {
                        if (lane == 0) {
                            const uint32_t at = atomicAdd(heavy_count, 1u);
                            heavy_rows[at] = row;
                            heavy_rows[n_left + at] = 0u;   // from its first visit
                        }
                        continue;
                    }
                }
            }
            // ... and a row that turns out to have MANY CANDIDATES hands the visits it has left to the launch over parts as
            // well: a call of the exact scoring is ~10 us per 64 candidates, a hub of a few thousand near-identical names is
            // 0.5 - 0.7 ms in one wave -- at 663 k the launch over an eighth's rows ended 0.4 ms after its median wave, on
            // such rows (scripts/wave_times_probe.py).  Known only while it happens, hence decided at the end of a visit:
            // the row keeps what it has found (its result row is written as usual, pass 2 merges the parts' matches into
            // it like mirrored ones) and the parts start at visit `v_end`.
            uint32_t row_scored = 0;      // pairs of this row scored so far (flush_s; kept in a vector register: the scalar ones are short)
            asm volatile("" : "+v"(row_scored));
            // Segment ends of the visits, four per 16-byte load: EA holds an even group of four visits, EB an odd one -- the
            // group in use and the next one, which is loaded BY HAND into the other buffer when the lane's stream enters a
            // group (load_ends below).  Two buffers that are never copied, because every way of writing "current = next;
            // next = load" ends in the compiler loading into a temporary and copying it into place behind a vmcnt(0)
            // right where the load is issued -- a full memory round trip AND the rounds in flight drained, every fourth
            // visit (round 3 lived with that: 1.2 M such stalls per launch at 663 k).
            u32x4 EA, EB;
            {
                const uint32_t g0 = v_lo >> 2;
                const uint4 a = ends8_at(g0), b = ends8_at(min(g0 + 1u, last_group8));
                const u32x4 av = u32x4{a.x, a.y, a.z, a.w}, bv = u32x4{b.x, b.y, b.z, b.w};
                EA = (g0 & 1u) ? bv : av;
                EB = (g0 & 1u) ? av : bv;
            }
            auto end_of_visit = [&](uint32_t v) {   // v is wave-uniform: a choice of register, not a computation
                const uint32_t c = v & 3u;
                const uint32_t ea = c == 0 ? EA.x : (c == 1 ? EA.y : (c == 2 ? EA.z : EA.w));
                const uint32_t eb = c == 0 ? EB.x : (c == 1 ? EB.y : (c == 2 ? EB.z : EB.w));
                return (v & 4u) ? eb : ea;
            };
            const uint32_t erow8_b = erow8 << 2;
            // sel: 0 = load group `grp` into EA, 1 = into EB, 2 = nothing to load (most rounds).  ONE asm statement that
            // the compiler sees on every path, with EA and EB as read-write operands: no branch of its own around it, so no
            // copy of a register that is still in flight.  Nobody waits for this load by name: loads return in order, it
            // is older than every round issued behind it, and it is first read four visits -- at least four rounds and
            // their waits -- later.
            auto load_ends = [&](uint32_t sel, uint32_t grp) {
                uint32_t e_off;
                asm volatile(
                    "s_cmp_gt_u32 %[sel], 1\n\t"
                    "s_cbranch_scc1 .Lends_done%=\n\t"
                    "v_lshl_add_u32 %[off], %[grp], 4, %[erow]\n\t"
                    "s_cmp_eq_u32 %[sel], 0\n\t"
                    "s_cbranch_scc0 .Lends_b%=\n\t"
                    "global_load_dwordx4 %[EA], %[off], %[base]\n\t"
                    "s_branch .Lends_done%=\n"
                    ".Lends_b%=:\n\t"
                    "global_load_dwordx4 %[EB], %[off], %[base]\n"
                    ".Lends_done%=:"
                    : [EA] "+v"(EA), [EB] "+v"(EB), [off] "=&v"(e_off)
                    : [sel] "s"(sel), [grp] "s"(grp), [erow] "v"(erow8_b), [base] "s"(ends8)
                    : "scc", "memory");
            };
            uint32_t gv = v_lo;                               // the visit whose loads are being issued (wave-uniform)
            uint32_t hi = min(end_of_visit(v_lo), hi_stop);   // end of the lane's segment in visit gv

            struct SBatch {
                u32x4 q;       // the lane's four entries of the round
            };
            const uint32_t null_at = null_off + ((uint32_t)lane << 4);   // this lane's four postings that add nothing
            bool live = hi > cur;                             // the lane's segment in visit gv holds entries at and behind `cur`
            // One round's load for every lane.  tv = the visit it belongs to (>= n_visits: past the end), last = the
            // visit ends with this round.
            auto issue_s = [&](SBatch &bt, uint32_t &tv, bool &last) {
                tv = gv;
                // Unconditional (see the tile-by-tile form) and WITHOUT validity masks: a lane that is through with its
                // segment reads four entries K3 leaves behind the array for it -- they add 0 to accumulators of their own -- and the
                // last load of a segment may bring one to three entries of what follows it (the term's next super-tile,
                // or the next term's list): whatever they are, they add non-negative amounts to accumulators of this
                // visit, which keeps every accumulator an UPPER bound (a few false positives more; the columns such
                // entries record are columns of this visit like any other and are scored exactly).  Four compares, four
                // selects and four mask operations per round less, and one register per round in flight.
                //
                // The load and its wait are written by hand.  With ordinary loads the compiler's own wait counts come out
                // as vmcnt(0) / vmcnt(1) at the head of every trip whatever the shape of the loop (its merge of the loop's
                // entry and back edge, profiles/r03_stream_waitcnt.md), i.e. the rounds in flight are drained once per trip
                // and the kernel runs at the latency of its loads (11.1 ms; the tile-by-tile form: 12.2).  An asm load is
                // invisible to that bookkeeping; apply_s waits for it with the exact count -- three younger rounds are in
                // flight whenever a round is applied; loads the compiler issues itself in between (segment ends) only make
                // the wait stricter, never too lax (loads return in order).  tests/test_kernel_isa.py checks that nothing
                // reads a round's registers between its load and its wait.
                {
                    const uint32_t at = live ? cur : null_at;
                    asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(bt.q) : "v"(at), "s"(filt) : "memory");
                }
                cur += G16;
                live = hi > cur;   // (one compare per round: the next round's choice of address is this one's "more to come")
                last = ballot64(live) == 0;
                uint32_t sel = 2u, grp = 0u;
                if (last) {   // next visit
                    ++gv;
                    if ((gv & 3u) == 0u) {   // a new group of four: the one behind it goes into the buffer that has just been left
                        grp = min((gv >> 2) + 1u, last_group8);
                        sel = ((gv >> 2) + 1u) & 1u;
                    }
                    cur = hi + u16;              // the next segment starts where this one ends
                    hi = min(end_of_visit(gv), hi_stop);
                    live = hi > cur;
                }
                load_ends(sel, grp);
            };
            SBatch sb0, sb1, sb2, sb3;
            sb0.q = sb1.q = sb2.q = sb3.q = u32x4{0u, 0u, 0u, 0u};   // (the loads name their registers as read-write operands)
            // A posting that finds its accumulator at or above its column's threshold records the column: appended to the
            // survivor buffer and nothing else -- at 663 k a round records five columns on average (every posting of a
            // column behind the first that passes fires again), so this path is not a rare one, and an LDS round trip per
            // firing slot (the exchange with the table of recorded columns) cost as much as the rest of the round
            // (profiles/r03_sessionE_*).  Repeats are removed in bulk when the buffer runs full (flush_s).
            uint32_t n_clean = 0;   // survivors at the front of the buffer that the repeat filter has seen already
            // (the rounds in flight land before the call: callees save and restore the registers they are loaded into, and
            //  the compiler does not know that they are in flight)
            // (oa, ob, oc: the three rounds in flight.  The round being applied has landed and is NOT named: naming it would
            //  make it a new value next to the copy the slots behind the call still read -- eight register moves per round
            //  that records anything, i.e. most rounds)
            auto flush_s = [&](SBatch &oa, SBatch &ob, SBatch &oc, uint32_t tv) {
                asm volatile("s_waitcnt vmcnt(0) ; rounds %0 %1 %2 ends %3 %4" : "+v"(oa.q), "+v"(ob.q), "+v"(oc.q), "+v"(EA), "+v"(EB)::"memory");
                const FlushOut<T> fo = flush_survivors<T, SYM, TILE_LOG2, WIDE>(nnz, thr, row_arg, sc, pairs, top, n_surv, n_clean);
                top = fo.top;
                const uint32_t fo_word = (uint32_t)__builtin_amdgcn_readfirstlane((int)fo.n_surv);   // left | handed to the scoring << 16
                st_surv += fo_word >> 16;
                if (CAN_SPLIT) {
                    // enough candidates for a row (pairs scored ~ the rounds' bar in time): the visit being applied is its
                    // last, the parts take over behind it (the rounds in flight of later visits are dropped by the loop's
                    // own end; once: tv + 3 > v_end from here on)
                    row_scored += fo_word >> 16;
                    if (!part_mode && part_cfg != 0u && tv + 3u <= v_end && ballot64((uint64_t)row_scored >= ((uint64_t)(part_cfg & 0x0fffffffu) << ((part_cfg >> 28) & 7u))) != 0) v_end = tv + 1u;   // (64 bits: 2^28 rounds << 7 does not fit 32)
                }
                n_surv = n_clean = fo_word & 0xffffu;
            };
            // `fired`: the lane's own test, `cm` its wave mask.  Written so that nothing of the bookkeeping goes through the
            // VALU that a scalar can do: the mask of the lanes that record is the AND of two compare masks (the ballot of a
            // COMBINED predicate is rebuilt through a select and a compare), a lane's slot in the buffer comes from mbcnt,
            // and the store runs under the two compares' masks.  16 -> 9 VALU per call, two to three calls per round
            // (profiles/r03_final_sq_counters.log: 128 VALU per round, 50 of them the round's own).
            auto collect_s = [&](bool fired, uint64_t cm, uint32_t r, uint32_t tv, SBatch &oa, SBatch &ob, SBatch &oc) {
                const uint32_t col = (tv << (TILE_LOG2 + FOLD_LOG2)) | ((r >> 1) & COL_MASK);   // bits [1, 16): fold, word, half
                // SYM: the pair (i, j > i) is row j's to score.  One-sided: entries read past the end of a segment (see
                // issue_s) may name columns of the last super-tile that do not exist.
                const bool mine = SYM ? col <= row : col < n_right;
                cm &= ballot64(mine);
                const uint32_t n_new = (uint32_t)__popcll(cm);
                if (n_surv + n_new > (uint32_t)SG_SURV_CAP - 2u) flush_s(oa, ob, oc, tv);   // (leaves fewer than 64; the buffer's last two words are not columns)
                const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(cm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)cm, 0u));
                if (fired && mine) surv[n_surv + below] = (int)col;
                n_surv += n_new;
            };
            struct SSlot {
                uint32_t z, sh, xs, x;
                uint32_t d;
            };
            // what the LDS add needs (address, amount): 4 VALU.  The posting goes into the 24-bit multiply AS IT IS -- its
            // bits below bq count as part of the value, which K3 has allowed for when it chose bq (emit_posting) ...
            auto prep_s = [&](uint32_t r) {
                SSlot q;
                q.z = r & ADDR_MASK;
                q.sh = r << 3;   // bit 4 = the half (bit 1 of the posting; bit 0 is 0)
                q.x = (uint32_t)(((uint64_t)(r & 0xffffffu) * (uint64_t)(CA & 0xffffffu)) >> 32);
                asm("v_lshlrev_b32 %0, %1, %2" : "=v"(q.xs) : "v"(q.sh), "v"(q.x));
                q.d = 0;
                return q;
            };
            // ... and what only the test of its result needs: the accumulator, with this posting's x added, must reach the
            // column's threshold -- ((old + x) << 16) >= d, d = T0s + C1n * int16(r >> 16): 1 VALU while the adds are on
            // their way, 3 behind them (bit field, add + shift, compare).
            auto bar_s = [&](uint32_t r) {
                uint32_t d;
                asm("v_mad_i32_i16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(d) : "v"(r), "v"(C1n), "v"(T0s));
                return d;
            };
            bool dirty = false;   // accumulators of the current visit hold sums
            auto apply_s = [&](SBatch &bt, SBatch &oa, SBatch &ob, SBatch &oc, uint32_t tv, bool last) {
                // the round's load is waited for HERE, on every path (also when no lane has a posting)
                asm volatile("s_waitcnt vmcnt(3) ; round %0" : "+v"(bt.q)::"memory");
                const u32x4 q = bt.q;
                {
                    dirty = true;
                    // every add leaves as soon as its address and amount are there; the thresholds of the four tests are
                    // computed behind the last add, i.e. inside the LDS round trip (the scheduler, left alone, computes
                    // all four slots first, then sends the four adds, and one threshold behind the wait -- which costs the
                    // same time: profiles/r03_sessionAA_slot_order_ab.log; kept because the ISA now reads as the source)
                    SSlot s0 = prep_s(q.x);
                    uint32_t o0 = __hip_atomic_fetch_add(tab_at(s0.z), s0.xs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    SG_SCHED_FENCE();
                    SSlot s1 = prep_s(q.y);
                    uint32_t o1 = __hip_atomic_fetch_add(tab_at(s1.z), s1.xs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    SG_SCHED_FENCE();
                    SSlot s2 = prep_s(q.z);
                    uint32_t o2 = __hip_atomic_fetch_add(tab_at(s2.z), s2.xs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    SG_SCHED_FENCE();
                    SSlot s3 = prep_s(q.w);
                    uint32_t o3 = __hip_atomic_fetch_add(tab_at(s3.z), s3.xs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    SG_SCHED_FENCE();
                    s0.d = bar_s(q.x);
                    s1.d = bar_s(q.y);
                    s2.d = bar_s(q.z);
                    s3.d = bar_s(q.w);
                    asm volatile("" : "+v"(s0.d), "+v"(s1.d), "+v"(s2.d), "+v"(s3.d));   // (before the wait, not behind it)
                    asm volatile("" : "+v"(o0), "+v"(o1), "+v"(o2), "+v"(o3));   // the one wait
                    // (old + x never leaves sixteen bits -- the accumulators are laid out for that -- and d >= 2^16: unsigned)
                    const bool f0 = ((__builtin_amdgcn_ubfe(o0, s0.sh, 16u) + s0.x) << 16) >= s0.d;
                    const bool f1 = ((__builtin_amdgcn_ubfe(o1, s1.sh, 16u) + s1.x) << 16) >= s1.d;
                    const bool f2 = ((__builtin_amdgcn_ubfe(o2, s2.sh, 16u) + s2.x) << 16) >= s2.d;
                    const bool f3 = ((__builtin_amdgcn_ubfe(o3, s3.sh, 16u) + s3.x) << 16) >= s3.d;
                    const uint64_t c0 = ballot64(f0), c1m = ballot64(f1), c2 = ballot64(f2), c3 = ballot64(f3);
                    if (c0 | c1m | c2 | c3) {
                        if (c0) collect_s(f0, c0, q.x, tv, oa, ob, oc);
                        if (c1m) collect_s(f1, c1m, q.y, tv, oa, ob, oc);
                        if (c2) collect_s(f2, c2, q.z, tv, oa, ob, oc);
                        if (c3) collect_s(f3, c3, q.w, tv, oa, ob, oc);
                    }
                }
                if (last && dirty) {   // the visit is through: its accumulators back to zero (eight stores at constant offsets)
#pragma unroll
                    for (int x = 0; x < TILE * 2 / 16; x += 64) tab_v[x + lane] = make_uint4(0, 0, 0, 0);
                    dirty = false;
                }
            };
            uint32_t tv0, tv1, tv2, tv3;
            bool la0, la1, la2, la3;
            issue_s(sb0, tv0, la0);
            issue_s(sb1, tv1, la1);
            issue_s(sb2, tv2, la2);
            issue_s(sb3, tv3, la3);
            SG_WD_DECL(wd_s);
            for (;;) {   // trips of four rounds; every round is re-issued right behind its use
                SG_WD(wd_s, 1 << 26, 17)
                if (tv0 >= v_end) break;
                apply_s(sb0, sb1, sb2, sb3, tv0, la0);
                issue_s(sb0, tv0, la0);
                if (tv1 >= v_end) break;
                apply_s(sb1, sb2, sb3, sb0, tv1, la1);
                issue_s(sb1, tv1, la1);
                if (tv2 >= v_end) break;
                apply_s(sb2, sb3, sb0, sb1, tv2, la2);
                issue_s(sb2, tv2, la2);
                if (tv3 >= v_end) break;
                apply_s(sb3, sb0, sb1, sb2, tv3, la3);
                issue_s(sb3, tv3, la3);
            }
            // rounds issued past the end of the stream are still in flight: they must land before their registers are reused
            asm volatile("s_waitcnt vmcnt(0) ; rounds %0 %1 %2 %3 ends %4 %5" : "+v"(sb0.q), "+v"(sb1.q), "+v"(sb2.q), "+v"(sb3.q), "+v"(EA), "+v"(EB)::"memory");
            if (n_surv > n_clean) flush_s(sb1, sb2, sb3, v_end);   // repeats out of what is left (scored below; everything has landed)
            if (CAN_SPLIT && !part_mode && v_end != n_visits && lane == 0) {   // handed on
                const uint32_t at = atomicAdd(heavy_count, 1u);
                heavy_rows[at] = row;
                heavy_rows[n_left + at] = v_end;
            }
        }
        {   // postings streamed = entries of P's lists in the tiles visited
            uint32_t mine = 0;
            if (g && u == 0 && part_lo == 0u) {   // (a row in parts: counted with its first part)
                const uint32_t *erp = ends + (size_t)erow;
                mine = (erp[t_end - 1u] >> 2) - my_lo;
            }
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) mine += __shfl_xor(mine, d, 64);
            st_post += (unsigned long long)wave_read<uint32_t>(mine, 0);
        }
        if (n_surv > 0) {   // fewer than 64 left
            top = drain_survivors<T, SYM, TILE_LOG2, WIDE, (FOLD_LOG2 > 0)>(nnz, thr, row_arg, sc, pairs, top, n_surv);
            st_surv += n_surv;
        }
        if (CAN_SPLIT && part_mode) {
            // a part's matches go to the pair list, addressed to the part's own row: pass 2 merges the parts like mirrored matches
            int cnt = __popcll(__ballot(top.c != INT32_MAX));
            if (cnt > keep) cnt = keep;
            emit_part_matches<T, TILE_LOG2>(pairs, top.s, top.c, cnt, row_out);
        } else {   // (symmetric mode: the row's matches j <= i; pass 2 merges the mirrored ones in)
            int cnt = __popcll(__ballot(top.c != INT32_MAX));
            if (cnt > keep) cnt = keep;
            const size_t obase = (size_t)row_out * (size_t)out_stride;
            if (lane < cnt) {
                out_vals[obase + lane] = top.s;
                out_cols[obase + lane] = top.c;
            }
            if (lane == 0) out_cnt[row_out] = cnt;
        }
        if (SYM) {
            // The pair list is full (this wave was handed a chunk past its end): whatever the pass still does is thrown away,
            // the caller runs the one-sided form.  Push the row counter past the last row so that every wave leaves at its
            // next helping.  (Looking at the shared chunk counter instead -- one load per helping -- cost 3.3 ms at 663 k:
            // accesses to one word are serialised at ~12 ns each, profiles/r02_sessionZ_abort_check_ab.log.)
            const uint32_t pos = (uint32_t)__builtin_amdgcn_readfirstlane(surv[SG_SURV_CAP - 1]);
            if (pos != SG_PAIR_NO_CHUNK && (pos >> 9) >= pair_chunks && lane == 0) atomicMax(row_counter, 0x20000000u);
        }
#ifdef SG_DEBUG_WAVE_TIMES
        {   // the wave's slowest row: {duration, row | pairs scored << 32}
            const unsigned long long dt = __builtin_amdgcn_s_memrealtime() - dbg_r0;
            if (dt > dbg_worst) {
                dbg_worst = dt;
                dbg_worst_row = (unsigned long long)row | ((st_surv - dbg_surv0) << 32);
            }
        }
#endif
    }
    }
    if (SYM && lane == 0) {   // close the wave's last chunk
        const uint32_t pos = (uint32_t)surv[SG_SURV_CAP - 1];
        if (pos != SG_PAIR_NO_CHUNK && (pos >> 9) < pair_chunks) {
            pairs->d_chunk_count[pos >> 9] = pos & 511u;
            atomicAdd(pairs->d_totals, (unsigned long long)(pos & 511u));
        }
    }
    if (lane == 0) {
        if (st_rows) atomicAdd(stats + 0, st_rows);
        if (st_post) atomicAdd(stats + 1, st_post);
        if (st_surv) atomicAdd(stats + 2, st_surv);
        // [4]: pairs scored exactly (what the second filter let through; without it: every survivor)
        if (FOLD_LOG2 > 0) {
            if (surv[SG_SURV_CAP - 2]) atomicAdd(stats + 4, (unsigned long long)(uint32_t)surv[SG_SURV_CAP - 2]);
        } else if (st_surv) atomicAdd(stats + 4, st_surv);
    }
#ifdef SG_DEBUG_WAVE_TIMES
    if (SYM && !WIDE && FOLD_LOG2 > 0 && sizeof(T) == 4 && lane == 0 && blockIdx.x < 8192u) {
        unsigned long long *w = sg_debug_wave_times + 4u * (blockIdx.x + (part_mode ? 8192u : 0u));
        w[0] = dbg_t0;
        w[1] = __builtin_amdgcn_s_memrealtime();
        w[2] = dbg_worst;
        w[3] = dbg_worst_row;
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// Symmetric mode, second pass: the pair list (i, j < i, s) -> for every row j the list of the rows i > j that match it
// -> merged with the row's own top list (its matches <= j, written by pass 1) -> top-n.
// (one workgroup per chunk of the pair list, one thread per entry)
// The statistics of the pass that counted ([0..2] rows / postings / pairs of pass 1, [3] rows that went through the exact
// kernel) copied into the context's words by the FIRST thread of a kernel of the second pass (two device-to-device
// copies of a few bytes were two launches of their own).
__device__ __forceinline__ void publish_pass_stats(const unsigned long long *src, const uint32_t *word, unsigned long long *dst) {
    if (dst != nullptr && blockIdx.x == 0 && threadIdx.x == 0) {
        dst[0] = src[0];
        dst[1] = src[1];
        dst[2] = src[2];
        dst[4] = src[4];
        reinterpret_cast<uint32_t *>(dst + 3)[0] = *word;
    }
}

template <typename T>
__global__ void __launch_bounds__(SG_PAIR_CHUNK) pairs_fill_kernel(const uint32_t *__restrict__ pi, const uint32_t *__restrict__ pj,
                                                                   const T *__restrict__ ps, const uint32_t *__restrict__ chunk_count,
                                                                   const uint32_t *__restrict__ ptr, uint32_t *cursor,
                                                                   int32_t *__restrict__ lcol, T *__restrict__ lval) {
    if (threadIdx.x >= chunk_count[blockIdx.x]) return;
    const size_t p = (size_t)blockIdx.x * SG_PAIR_CHUNK + threadIdx.x;
    const uint32_t j = pj[p];
    const uint32_t at = ptr[j] + atomicAdd(&cursor[j], 1u);
    lcol[at] = (int32_t)pi[p];
    lval[at] = ps[p];
}

// ---- multi-GPU form of the second pass: the pair list leaves the GPU as one array of {i, j, score bits} records
// (W = 3 words for f32, 4 for f64), the lists of all ranks come back concatenated, and every rank merges the pairs
// whose row j lies in its range.
template <typename T>
__global__ void __launch_bounds__(SG_PAIR_CHUNK) pairs_export_kernel(const uint32_t *__restrict__ pi, const uint32_t *__restrict__ pj,
                                                                     const T *__restrict__ ps, const uint32_t *__restrict__ chunk_count,
                                                                     const uint32_t *__restrict__ chunk_start, int32_t *__restrict__ out,
                                                                     const unsigned long long *st_src, const uint32_t *st_word,
                                                                     unsigned long long *st_dst) {
    constexpr int W = sizeof(T) == 8 ? 4 : 3;
    publish_pass_stats(st_src, st_word, st_dst);
    if (threadIdx.x >= chunk_count[blockIdx.x]) return;
    const size_t p = (size_t)blockIdx.x * SG_PAIR_CHUNK + threadIdx.x;
    int32_t *o = out + ((size_t)chunk_start[blockIdx.x] + threadIdx.x) * W;
    o[0] = (int32_t)pi[p];
    o[1] = (int32_t)pj[p];
    if (sizeof(T) == 8) {
        const long long b = __double_as_longlong((double)ps[p]);
        o[2] = (int32_t)(b & 0xffffffffll);
        o[3] = (int32_t)(b >> 32);
    } else {
        o[2] = __float_as_int((float)ps[p]);
    }
}

template <int W>
__global__ void __launch_bounds__(256) pairs_flat_count_kernel(const int32_t *__restrict__ pairs, int64_t n_pairs, uint32_t lo,
                                                               uint32_t hi, uint32_t step, const uint32_t *__restrict__ pos_of,
                                                               uint32_t *cnt) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pairs) return;
    const uint32_t j = (uint32_t)pairs[p * W + 1];
    const uint32_t at = pos_of ? pos_of[j] : j;   // a rank's range is a range of POSITIONS when the index is permuted
    if (at >= lo && at < hi && (hi - 1u - at) % step == 0u) atomicAdd(&cnt[j], 1u);   // (a rank's rows: hi - 1, hi - 1 - step, ...)
}

template <typename T>
__global__ void __launch_bounds__(256) pairs_flat_fill_kernel(const int32_t *__restrict__ pairs, int64_t n_pairs, uint32_t lo,
                                                              uint32_t hi, uint32_t step, const uint32_t *__restrict__ pos_of,
                                                              const uint32_t *__restrict__ ptr, uint32_t *cursor,
                                                              int32_t *__restrict__ lcol, T *__restrict__ lval) {
    constexpr int W = sizeof(T) == 8 ? 4 : 3;
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pairs) return;
    const int32_t *rec = pairs + p * W;
    const uint32_t j = (uint32_t)rec[1];
    const uint32_t pj = pos_of ? pos_of[j] : j;
    if (pj < lo || pj >= hi || (hi - 1u - pj) % step != 0u) return;
    const uint32_t at = ptr[j] + atomicAdd(&cursor[j], 1u);
    lcol[at] = rec[0];
    if (sizeof(T) == 8)
        lval[at] = (T)__longlong_as_double(((long long)rec[3] << 32) | (long long)(uint32_t)rec[2]);
    else
        lval[at] = (T)__int_as_float(rec[2]);
}

// A wave looks at 64 rows at a time and works on those that have mirrored matches: own list + mirrored list through the
// rank sort (up to 64 candidates, the common case) or the register top-n list (order of arrival is irrelevant).
template <typename T>
__global__ void __launch_bounds__(64) pairs_select_kernel(const uint32_t *__restrict__ ptr, const int32_t *__restrict__ lcol,
                                                          const T *__restrict__ lval, uint32_t n_rows, int32_t keep,
                                                          int32_t out_stride, int32_t *__restrict__ out_cols,
                                                          T *__restrict__ out_vals, int32_t *__restrict__ out_cnt,
                                                          const unsigned long long *st_src, const uint32_t *st_word,
                                                          unsigned long long *st_dst) {
    const int lane = threadIdx.x;
    publish_pass_stats(st_src, st_word, st_dst);
    // The 64 rows a wave looks at are n_groups apart, not neighbours: on a sorted list the rows with long mirrored lists
    // (hubs of near-identical names) ARE neighbours, and a wave that owned 64 of them in a row worked through them one
    // after the other while the others idled (4.9 ms instead of 0.25 at 663 k sorted names).
    const uint32_t n_groups = (n_rows + 63u) / 64u;
    for (uint32_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
        const uint32_t mine = g + (uint32_t)lane * n_groups;
        uint64_t todo = __ballot(mine < n_rows && ptr[mine + 1] != ptr[mine]);
        while (todo) {
            const uint32_t row = g + (uint32_t)__builtin_ctzll(todo) * n_groups;
            todo &= todo - 1;
            const uint32_t lo = ptr[row], hi = ptr[row + 1];
            const int own = out_cnt[row];
            const size_t obase = (size_t)row * (size_t)out_stride;
            const int m = own + (int)(hi - lo);
            if (m <= 64) {
                // every lane holds one candidate and ranks it against the others
                const bool have = lane < m;
                T s = (T)-INFINITY;
                int c = INT32_MAX;
                if (lane < own) {
                    s = out_vals[obase + lane];
                    c = out_cols[obase + lane];
                } else if (have) {
                    s = lval[lo + (uint32_t)(lane - own)];
                    c = lcol[lo + (uint32_t)(lane - own)];
                }
                int rank = 0;
                bool repeat = false;   // the same pair earlier in the list (the stream form may score a pair twice)
                for (int q = 0; q < m; ++q) {
                    const T sq = wave_read<T>(s, q);
                    const int cq = wave_read<int>(c, q);
                    rank += (sq > s || (sq == s && cq < c)) ? 1 : 0;
                    repeat = repeat || (cq == c && q < lane);
                }
                uint64_t rm = __ballot(have && repeat);
                const int distinct = m - __popcll(rm);
                while (rm) {   // (rare) a repeat ranked ahead of this entry does not count
                    const int q = __builtin_ctzll(rm);
                    rm &= rm - 1;
                    const T sq = wave_read<T>(s, q);
                    const int cq = wave_read<int>(c, q);
                    rank -= (sq > s || (sq == s && cq < c)) ? 1 : 0;
                }
                __builtin_amdgcn_wave_barrier();   // all of the own list is in registers before it is overwritten
                if (have && !repeat && rank < keep) {
                    out_vals[obase + rank] = s;
                    out_cols[obase + rank] = c;
                }
                if (lane == 0) out_cnt[row] = distinct < keep ? distinct : keep;
                continue;
            }
            if (keep > SG_TOPN_LANES) {   // top_n of 65 .. 128: two register lists (the pass has sent the row's own matches here as well)
                TopListWide<T> wide;
                wide.clear();
                for (int base = 0; base < own; base += 64) {
                    const bool have = base + lane < own;
                    const T s = have ? out_vals[obase + base + lane] : (T)0;
                    const int c = have ? out_cols[obase + base + lane] : 0;
                    const int mm = min(64, own - base);
                    for (int q = 0; q < mm; ++q) wide.insert_unique(wave_read<T>(s, q), wave_read<int>(c, q), lane);
                }
                for (uint32_t base = lo; base < hi; base += 64) {
                    const bool have = base + (uint32_t)lane < hi;
                    const T s = have ? lval[base + lane] : (T)0;
                    const int c = have ? lcol[base + lane] : 0;
                    // (only what can still make the row's best `keep`: see the narrow list below)
                    const T bar_s = keep <= 64 ? wave_read<T>(wide.lo.s, keep - 1) : wave_read<T>(wide.hi.s, keep - 65);
                    const int bar_c = keep <= 64 ? wave_read<int>(wide.lo.c, keep - 1) : wave_read<int>(wide.hi.c, keep - 65);
                    uint64_t todo = __ballot(have && (s > bar_s || (s == bar_s && c < bar_c)));
                    while (todo) {
                        const int q = __builtin_ctzll(todo);
                        todo &= todo - 1;
                        wide.insert_unique(wave_read<T>(s, q), wave_read<int>(c, q), lane);
                    }
                }
                int cnt = wide.count();
                if (cnt > keep) cnt = keep;
                if (lane < cnt) {
                    out_vals[obase + lane] = wide.lo.s;
                    out_cols[obase + lane] = wide.lo.c;
                }
                if (lane + 64 < cnt) {
                    out_vals[obase + 64 + lane] = wide.hi.s;
                    out_cols[obase + 64 + lane] = wide.hi.c;
                }
                if (lane == 0) out_cnt[row] = cnt;
                continue;
            }
            TopList<T> top;
            top.clear();
            {
                const T s = lane < own ? out_vals[obase + lane] : (T)0;
                const int c = lane < own ? out_cols[obase + lane] : 0;
                for (int q = 0; q < own; ++q) top.insert(wave_read<T>(s, q), wave_read<int>(c, q), lane);
            }
            for (uint32_t base = lo; base < hi; base += 64) {
                const bool have = base + (uint32_t)lane < hi;
                const T s = have ? lval[base + lane] : (T)0;
                const int c = have ? lcol[base + lane] : 0;
                // Only what can still make the row's best `keep` is inserted: the list's keep-th entry is the bar (an empty slot
                // is (-inf, INT_MAX): everything passes until `keep` are there).  The bar only rises while a batch is worked
                // off, so the one read in front of it lets too much through, never too little.  At thresholds below the
                // name-matching range a row receives dozens of pairs for a top_n of ten: 200 k names at 0.3, the second pass
                // 4.0 -> 2.0 ms.
                const T bar_s = wave_read<T>(top.s, keep - 1);
                const int bar_c = wave_read<int>(top.c, keep - 1);
                uint64_t todo = __ballot(have && (s > bar_s || (s == bar_s && c < bar_c)));
                while (todo) {
                    const int q = __builtin_ctzll(todo);
                    todo &= todo - 1;
                    top.insert_unique(wave_read<T>(s, q), wave_read<int>(c, q), lane);
                }
            }
            int cnt = __popcll(__ballot(top.c != INT32_MAX));
            if (cnt > keep) cnt = keep;
            if (lane < cnt) {
                out_vals[obase + lane] = top.s;
                out_cols[obase + lane] = top.c;
            }
            if (lane == 0) out_cnt[row] = cnt;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Properties a matrix must have for the pruned kernel: values >= 0 (no NaN), column indices strictly
// ascending within every row, row norms <= 1 (+ rounding).  One thread per row.
template <typename T>
__global__ void __launch_bounds__(256) csr_props_kernel(const int64_t *__restrict__ indptr,
                                                        const int32_t *__restrict__ indices,
                                                        const T *__restrict__ data, int64_t n_rows,
                                                        uint32_t *out /* [0] violations [1] max ||row||^2 as float bits [2] longest row */) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t bad = 0, len = 0;
    float n2 = 0.f;
    if (i < n_rows) {
        len = (uint32_t)(indptr[i + 1] - indptr[i]);
        double s = 0.0;
        int prev = -1;
        for (int64_t p = indptr[i]; p < indptr[i + 1]; ++p) {
            const T v = data[p];
            const int k = indices[p];
            if (!(v >= (T)0)) bad = 1;
            if (k <= prev) bad = 1;
            prev = k;
            s += (double)v * (double)v;
        }
        n2 = __double2float_ru(s);
    }
    uint32_t nb = __float_as_uint(n2);   // non-negative floats order like unsigned integers
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        nb = max(nb, (uint32_t)__shfl_xor((int)nb, d, 64));
        len = max(len, (uint32_t)__shfl_xor((int)len, d, 64));
        bad |= (uint32_t)__shfl_xor((int)bad, d, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        if (bad) atomicOr(out, 1u);
        atomicMax(out + 1, nb);
        atomicMax(out + 2, len);
    }
}

int sg_csr_props(sg_ctx *ctx, const sg_csr *m, bool *cosine_like, float *max_norm2, uint32_t *max_nnz) {
    // A matrix made by the vectoriser (K2) is cosine-like BY CONSTRUCTION -- counts times positive weights, columns in key
    // order, every row divided by its norm -- and its largest squared row norm is 1 up to the rounding of the division
    // (two roundings per entry: <= 1 + 3 * 2^-24 in fp32): nothing has to come back from the device for that (round 4:
    // the read-back was one of the step's ten synchronisations).  Only the longest row is not known this way; the one
    // caller that wants it (the opt-in row blocks, SG_ROW_BLOCKS=1) still reads the words K2 leaves.
    if (m->props_state == 0 && m->from_vectoriser && max_nnz == nullptr) {
        m->props_max_norm2 = 1.000001f;
        m->props_max_nnz = 0;
        m->props_state = 1;
        m->props_by_construction = true;
    }
    if (m->props_state == 0 || (max_nnz != nullptr && m->props_by_construction && m->props_max_nnz == 0)) {
        uint32_t *d = nullptr;
        SG_TRY(sg_alloc(ctx, (size_t)4, &d));
        uint32_t h[4] = {0, 0, 0, 0};
        hipError_t e = hipMemsetAsync(d, 0, 16, ctx->stream);
        if (e == hipSuccess && m->d_props_words) {   // left by the vectoriser (K2): nothing to scan
            e = hipMemcpyAsync(d, m->d_props_words, 12, hipMemcpyDeviceToDevice, ctx->stream);
        } else if (e == hipSuccess && m->n_rows > 0) {
            const unsigned grid = (unsigned)((m->n_rows + 255) / 256);
            if (m->dtype == SG_F64)
                hipLaunchKernelGGL(csr_props_kernel<double>, dim3(grid), dim3(256), 0, ctx->stream, m->d_indptr,
                                   m->d_indices, (const double *)m->d_data, m->n_rows, d);
            else
                hipLaunchKernelGGL(csr_props_kernel<float>, dim3(grid), dim3(256), 0, ctx->stream, m->d_indptr,
                                   m->d_indices, (const float *)m->d_data, m->n_rows, d);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(h, d, 16, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        ctx->release(d);
        if (e != hipSuccess) {
            sg_set_error("sg_csr_props: %s", hipGetErrorString(e));
            return SG_ERR_HIP;
        }
        float n2;
        memcpy(&n2, &h[1], 4);
        if (!m->props_by_construction) m->props_max_norm2 = n2;     // (a bound already in use stays: the index may be built on it)
        m->props_max_nnz = h[2];
        m->props_state = (h[0] == 0 && n2 <= 1.0001f) ? 1 : 2;
    }
    *cosine_like = m->props_state == 1;
    *max_norm2 = m->props_max_norm2;
    if (max_nnz) *max_nnz = m->props_max_nnz;
    return SG_OK;
}

// ------------------------------------------------------------------------------------------------
struct PairList {   // symmetric mode: the mirrored pairs (i, j < i) above the threshold, in chunks of SG_PAIR_CHUNK entries
    uint32_t *d_i = nullptr;
    uint32_t *d_j = nullptr;
    void *d_s = nullptr;
    uint32_t *d_row_count = nullptr;           // mirrored matches per row (counted by pass 1)
    uint32_t *d_chunk_count = nullptr;         // entries used of every chunk
    uint32_t *d_chunks_used = nullptr;         // chunks handed out
    unsigned long long *d_totals = nullptr;    // pairs
    uint32_t chunks = 0;
    uint32_t row_lo = 0, row_hi = 0;           // the left rows to score (the whole matrix on one GPU):
    uint32_t row_step = 1;                     // row_hi - 1, row_hi - 1 - row_step, ... >= row_lo
    uint32_t rows() const { return (row_hi - row_lo + row_step - 1u) / row_step; }
    const SgPairSink *d_sink = nullptr;        // the same pointers as a struct in device memory: what the kernel is handed
};

__global__ void pair_sink_kernel(SgPairSink v, SgPairSink *out) { *out = v; }

// single-wave workgroups of the pruned kernel: as many as the LDS of the chip holds
// LDS of one wave: accumulator tile, row hash (keys 512 B, values up to 1 KiB), survivor buffer; the stream form adds its
// 512-byte table of recorded columns, which for f32 fits the unused half of the value slots
// The second filter per call: built with the index (name-length rows, sg_postings.hip), used in the stream form, and only at
// thresholds where most candidates of the first filter are false alarms -- at 0.6 a fifth of them are matches or close, the
// filter's pass over their entries is then work done twice (200 k names, top 20 at 0.6: 5.5 ms without, 6.0 with;
// at 0.7 and above it pays: 10 M x 1 M at 0.7 - 20 %).  SG_Q8_MIN_THRESHOLD moves the bar.
bool sg_q8_applies(const sg_ctx *ctx, const sg_postings *Bt, double threshold) {
    if (!Bt->d_q8 || Bt->fold_log2 <= 0) return false;
    double bar = 0.65;
    if (const char *v = ctx->opt("SG_Q8_MIN_THRESHOLD")) bar = atof(v);
    return threshold >= bar;
}

static size_t pruned_lds(int32_t tile_log2, int32_t fold_log2, int32_t dtype) {
    return ((size_t)2 << tile_log2) + 512 + 1024 + (size_t)SG_SURV_CAP * 4 + (fold_log2 > 0 && dtype == SG_F64 ? 512 : 0);
}
static unsigned pruned_grid(const sg_ctx *ctx, int32_t tile_log2, int64_t n_rows, int32_t fold_log2 = 0, int32_t dtype = SG_F32) {
    const size_t lds = pruned_lds(tile_log2, fold_log2, dtype);
    int waves_per_cu = (int)(ctx->lds_per_cu / lds);
    if (waves_per_cu > 32) waves_per_cu = 32;
    if (waves_per_cu < 1) waves_per_cu = 1;
    if (const char *v = ctx->opt("SG_PRUNE_WAVES_PER_CU"))
        if (atoi(v) > 0) waves_per_cu = atoi(v);
    unsigned grid = (unsigned)ctx->num_cu * (unsigned)waves_per_cu;
    if ((int64_t)grid > n_rows) grid = (unsigned)(n_rows > 0 ? n_rows : 1);
    return grid;
}

template <typename T, int TILE_LOG2, bool SYM, bool WIDE, int FOLD_LOG2>
static int launch_pruned(sg_ctx *ctx, const sg_csr *A, const sg_postings *Bt, int32_t keep, sg_topn *r, T thr,
                         float s_budget, uint32_t *row_counter, uint32_t *flagged_count, uint32_t *flagged_rows,
                         unsigned long long *stats, const PairList &pl, const uint32_t *row_list, const uint32_t *row_list_len,
                         uint32_t *heavy_count = nullptr, uint32_t *heavy_rows = nullptr, uint32_t part_cfg = 0) {
    const size_t lds = pruned_lds(TILE_LOG2, FOLD_LOG2, A->dtype);
    unsigned grid = pruned_grid(ctx, TILE_LOG2, SYM ? (int64_t)pl.rows() : A->n_rows, FOLD_LOG2, A->dtype);
    // the wide launch: few rows, if any, in a list of names (idle waves leave at once) -- unless the rows are long on average
    // (about 50 entries and up: a good part of them beyond 64), where a quarter of the chip was all it got (50 k strings of 100
    // entries: profiles/r06b_long_strings.log)
    if (WIDE && grid > (unsigned)ctx->num_cu * 4u && (double)A->nnz <= 48.0 * (double)A->n_rows) grid = (unsigned)ctx->num_cu * 4u;
    // (SHARE: see the kernel; only the forms that can run in parts have the second instantiation)
    constexpr bool SPLITS = SYM && !WIDE && FOLD_LOG2 > 0;
    const bool share = SPLITS && (heavy_count != nullptr || part_cfg != 0u);
    auto kernel = share ? spgemm_topn_pruned_kernel<T, TILE_LOG2, SYM, WIDE, FOLD_LOG2, SPLITS>
                        : spgemm_topn_pruned_kernel<T, TILE_LOG2, SYM, WIDE, FOLD_LOG2, false>;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(64), lds, ctx->stream, A->d_indptr,
                       A->d_indices, (const T *)A->d_data, (uint32_t)A->n_rows, (const uint32_t *)Bt->d_seg,
                       (const uint32_t *)Bt->d_ends, Bt->nt_pad, (uint32_t)Bt->n_terms,
                       (const uint32_t *)Bt->d_filt, Bt->n_tiles,
                       (const SgScoreCtx *)Bt->d_score_ctx + (sg_q8_applies(ctx, Bt, (double)thr) ? 0 : 1), keep, r->stride, thr, s_budget,
                       Bt->norm_up, Bt->freq_min, r->d_cols,
                       (T *)r->d_vals,
                       r->d_counts, row_counter, flagged_count, flagged_rows, stats, pl.d_sink, pl.chunks, pl.row_lo, pl.row_hi, row_list,
                       row_list_len, (const uint32_t *)Bt->d_ends8, Bt->nv_pad, (uint32_t)(Bt->nnz * 4), (uint32_t)Bt->n_right,
                       heavy_count, heavy_rows, pl.row_step, part_cfg);
    SG_HIP_TRY(hipGetLastError());
    return SG_OK;
}

// Both launches of one multiply: every row through the 64-term kernel; the rows it passes on (65 .. 128 non-zeros)
// through the wide one; what THAT passes on (more than 128 non-zeros, more than 64 prefix terms, delta too small for
// the fixed point) lands in (flagged_count, flagged_rows) for the exact kernel.
template <typename T, int TILE_LOG2, bool SYM, int FOLD_LOG2>
static int launch_both(sg_ctx *ctx, const sg_csr *A, const sg_postings *Bt, int32_t keep, sg_topn *r, T thr,
                       float s_budget, uint32_t *row_counter, uint32_t *flagged_count, uint32_t *flagged_rows,
                       unsigned long long *stats, const PairList &pl) {
    uint32_t *l1 = nullptr;
    SG_TRY(sg_alloc(ctx, (size_t)A->n_rows + 8, &l1));
    int st = hipMemsetAsync(l1, 0, 4 * sizeof(uint32_t), ctx->stream) == hipSuccess ? SG_OK : SG_ERR_HIP;
    // Stream + self-join form: rows that need many rounds -- a chain of dependent loads in one wave, milliseconds for the
    // slowest -- are set aside by the first launch and worked off in SG_ROW_PARTS parts each by a launch of their own
    // (see the kernel).  The whole-matrix pass hides such rows behind the others; a rank's RANGE of the multi-GPU form
    // ends with its slowest row (scripts/range_probe.py, scripts/sim_scaling.py).  SG_HEAVY_ROUNDS: rounds from which a
    // row is set aside (0: never).
    uint32_t *heavy = nullptr;   // [0] count [1] the parts' row counter [4 ..) rows
    uint32_t heavy_rounds = 0;
    if (SYM && FOLD_LOG2 > 0 && st == SG_OK) {
        // (the whole matrix in one pass: 6.0 ms without, 6.35 with the second launch at 663 k -- it has a ramp and a tail of
        //  its own; a range of an eighth: 3.2 -> 2.25 ms, profiles/r03_sessionS_*)
        // ... and a launch over at least half the rows still hides them better than parts do: a half share of the 663 k job
        // takes 3.6 ms without parts (the longest row: 3.4), 4.06 with; a quarter 3.2 without, 2.8 with
        // (profiles/r03_sessionAU_shares_without_parts.log)
        if ((pl.row_lo > 0 || (int64_t)pl.row_hi < A->n_rows || pl.row_step > 1) && 2 * (int64_t)pl.rows() < A->n_rows) {
            // A row is worth parts when it is a noticeable share of what ONE wave of the range does.  Rounds per wave,
            // estimated: rows per wave x rounds per row at the range's position (a row's stream grows with its position
            // and with the lists, i.e. with n: 40 rounds per row on average at 553 k index rows, 290 at 3.9 M --
            // 7.2e-5 n).  A fixed bar of 256 rounds sent nearly every row of the 5 M job through parts: eight ranges took
            // 278 ms in all against 204 ms for the whole (profiles/r03_sessionV_sim_scaling_5M.log).
            const double n_idx = (double)A->n_rows;
            const double rows_per_wave = (double)pl.rows() / (double)pruned_grid(ctx, TILE_LOG2, (int64_t)pl.rows(), FOLD_LOG2, A->dtype);
            const double rounds_per_row = 2.0 * 7.2e-5 * n_idx * (0.5 * ((double)pl.row_lo + (double)pl.row_hi) / n_idx);
            // (a twentieth: the tail a row of `bar` rounds can leave is then ~5 % of the launch's time.  A quarter was
            //  tried twice -- for contiguous ranges and for interleaved shares: shares of the 5 M job then ended 10-15 ms
            //  after the others, on single rows.  Also tried and dropped, profiles/r03_sessionAR_parts_first.log: a classify
            //  launch that lists such rows BEFORE the multiply, whose launch then starts with their parts -- 4.5 instead
            //  of 4.15 ms for a half share at 663 k, 1.94 instead of 1.78 for an eighth)
            double bar_share = 0.05;
            if (const char *v = ctx->opt("SG_HEAVY_SHARE")) bar_share = atof(v) > 0.0 ? atof(v) : bar_share;
            const double bar = bar_share * rows_per_wave * rounds_per_row;
            // (never below 128 rounds: at 663 k, eight ranges, bars of 64 / 128 / 256 / 512 rounds give 2.10 / 2.03 / 2.19 /
            //  2.57 ms for the slowest range -- profiles/r03_sessionAG_heavy_bar_ranges.log)
            heavy_rounds = bar < 128.0 ? 128u : (bar > 1.0e9 ? 1000000000u : (uint32_t)bar);
        }
        if (const char *v = ctx->opt("SG_HEAVY_ROUNDS")) heavy_rounds = (uint32_t)atoi(v) & 0x7fffffffu;
        if (heavy_rounds > 0x0fffffffu) heavy_rounds = 0x0fffffffu;
        // candidates per round of the bar from which a row hands its remaining visits to the parts (2^shift): about a
        // thousand candidates at the floor of the bar (128 rounds: the 663 k job), two per round of a large bar -- with the
        // second filter a candidate costs a fifth of what it did, and what keeps a wave late is a HUB's candidates, which
        // pass it and are scored exactly (5 M names, 8 shares, scripts/share_knob_sweep.sh: slowest share 23.2 -> 21.4 ms
        // with 2 instead of 8 per round; the 663 k shares lose 0.1 of 1.5 ms that way).  SG_HANDOVER_SHIFT overrides.
        uint32_t handover_shift = heavy_rounds >= 512u ? 1u : (heavy_rounds >= 256u ? 2u : 3u);
        if (const char *v = ctx->opt("SG_HANDOVER_SHIFT")) handover_shift = (uint32_t)atoi(v) & 7u;
        if (heavy_rounds) heavy_rounds |= handover_shift << 28;
        if (heavy_rounds) {
            st = sg_alloc(ctx, 2 * (size_t)A->n_rows + 8, &heavy);   // [4, 4 + n): the rows, [4 + n, 4 + 2 n): their first visit for the parts
            if (st == SG_OK && hipMemsetAsync(heavy, 0, 4 * sizeof(uint32_t), ctx->stream) != hipSuccess) st = SG_ERR_HIP;
        }
    }
    if (st == SG_OK)
        st = launch_pruned<T, TILE_LOG2, SYM, false, FOLD_LOG2>(ctx, A, Bt, keep, r, thr, s_budget, row_counter, l1, l1 + 4, stats, pl,
                                                     nullptr, nullptr, heavy, heavy ? heavy + 4 : nullptr, heavy ? heavy_rounds : 0u);
    if (st == SG_OK && heavy)
        st = launch_pruned<T, TILE_LOG2, SYM, false, FOLD_LOG2>(ctx, A, Bt, keep, r, thr, s_budget, heavy + 1, l1, l1 + 4, stats, pl,
                                                     heavy + 4, heavy, nullptr, nullptr, 0x80000000u);
    if (heavy) ctx->release(heavy);   // stream-ordered, like l1 below
    if (st == SG_OK && !(ctx->opt("SG_PRUNE_WIDE") && ctx->opt("SG_PRUNE_WIDE")[0] == '0'))
        st = launch_pruned<T, TILE_LOG2, SYM, true, FOLD_LOG2>(ctx, A, Bt, keep, r, thr, s_budget, l1 + 1, flagged_count, flagged_rows, stats,
                                                    pl, l1 + 4, l1);
    else if (st == SG_OK) {   // SG_PRUNE_WIDE=0: the first launch's list goes to the exact kernel as it is
        if (hipMemcpyAsync(flagged_count, l1, 4, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess ||
            hipMemcpyAsync(flagged_rows, l1 + 4, sizeof(uint32_t) * (size_t)A->n_rows, hipMemcpyDeviceToDevice, ctx->stream) !=
                hipSuccess)
            st = SG_ERR_HIP;
    }
    ctx->release(l1);   // stream-ordered: the pool hands it out again only to work queued behind these launches
    return st;
}

template <typename T, bool SYM>
static int dispatch_pruned(sg_ctx *ctx, const sg_csr *A, const sg_postings *Bt, int32_t keep, sg_topn *r, T thr,
                           float s_budget, uint32_t *row_counter, uint32_t *flagged_count, uint32_t *flagged_rows,
                           unsigned long long *stats, const PairList &pl) {
    if (Bt->fold_log2 > 0) {   // stream form: the postings were written for it (sg_postings.hip)
        if (Bt->fold_log2 != 3 || Bt->tile_log2 != 12 || !Bt->d_ends8) {
            sg_set_error("postings folded 2^%d over tiles of 2^%d columns are not supported by the pruned multiply", Bt->fold_log2,
                         Bt->tile_log2);
            return SG_ERR_UNSUPPORTED;
        }
        return launch_both<T, 12, SYM, 3>(ctx, A, Bt, keep, r, thr, s_budget, row_counter, flagged_count, flagged_rows, stats, pl);
    }
    switch (Bt->tile_log2) {
        case 11: return launch_both<T, 11, SYM, 0>(ctx, A, Bt, keep, r, thr, s_budget, row_counter, flagged_count, flagged_rows, stats, pl);
        case 12: return launch_both<T, 12, SYM, 0>(ctx, A, Bt, keep, r, thr, s_budget, row_counter, flagged_count, flagged_rows, stats, pl);
        case 13: return launch_both<T, 13, SYM, 0>(ctx, A, Bt, keep, r, thr, s_budget, row_counter, flagged_count, flagged_rows, stats, pl);
        default:
            sg_set_error("postings tile of 2^%d columns is not supported by the pruned multiply (2^11..2^13)", Bt->tile_log2);
            return SG_ERR_UNSUPPORTED;
    }
}

bool sg_pruned_supports_tile(int32_t tile_log2) { return tile_log2 >= 11 && tile_log2 <= 13; }

static float prune_budget(const sg_postings *Bt, double threshold, double delta) {
    // beta = threshold - delta bounds ||a_S|| * max ||b_j||; the kernel compares sums of squares of a
    const double nb = (double)Bt->norm_up;
    const double beta = threshold - delta;
    const double budget = (beta / nb) * (beta / nb) * (1.0 - 1e-6);
    return __builtin_nextafterf((float)budget, 0.f);
}

int sg_spgemm_pruned_launch(sg_ctx *ctx, const sg_csr *A, const sg_postings *Bt, int32_t keep, sg_topn *r,
                            double threshold, double delta, uint32_t *row_counter, uint32_t *flagged_count,
                            uint32_t *flagged_rows, unsigned long long *stats) {
    const float s_budget = prune_budget(Bt, threshold, delta);
    PairList none;
    if (A->dtype == SG_F64)
        return dispatch_pruned<double, false>(ctx, A, Bt, keep, r, (double)threshold, s_budget, row_counter, flagged_count,
                                              flagged_rows, stats, none);
    return dispatch_pruned<float, false>(ctx, A, Bt, keep, r, (float)threshold, s_budget, row_counter, flagged_count,
                                         flagged_rows, stats, none);
}

// Self-join form: pass 1 (the pruned kernel over the pairs j <= i, then the exact kernel's self-join launch over the
// rows the pruned kernel passed on) + the decision whether the pair list is complete (one host round trip: pair
// count, chunks handed out) + pass 2 (lists, top-n).
// *done == false: nothing usable was produced (too many pairs for the list) and the caller runs the one-sided
// form; the statistics words are untouched then (the result rows are overwritten by that form).
// exact_all: every row is handed to the exact kernel's self-join launch, from the last position down (a row's cost grows
// with its position) -- thresholds below the pruned kernel's envelope, or products its pilot prices dearer than the exact
// multiply: half the (row, tile) visits of the one-sided exact kernel, the same pair list and second pass.
// rows[i] = n - 1 - first - i for i < count; *len = count (what the launch reads), *listed = first + count (the statistics)
__global__ void __launch_bounds__(256) all_rows_descending_kernel(uint32_t n, uint32_t first, uint32_t count, uint32_t *__restrict__ rows,
                                                                  uint32_t *len, uint32_t *listed) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) rows[i] = n - 1u - first - i;
    if (i == 0) {
        *len = count;
        *listed = first + count;
    }
}

int sg_spgemm_pruned_symmetric(sg_ctx *ctx, const sg_csr *A, const sg_postings *Bt, int32_t keep, sg_topn *r,
                               double threshold, double delta, unsigned long long *stats, bool *done, int64_t row_lo,
                               int64_t row_hi, int32_t **export_pairs, int64_t *export_n, int64_t row_step, bool exact_all) {
    *done = false;
    if (A->n_rows >= ((int64_t)1 << 30)) return SG_OK;   // (positions carry two flag bits on their way to the survivor routine)
    // the self-join runs in position space: its left matrix is the one the index was built over (sg_postings.hip)
    if (Bt->permuted) A = Bt->permuted;
    else SG_TRY(sg_csr_ensure_rows(ctx, A));   // (no copy in position order: the rows of A itself are read)
    const size_t vs = A->dtype == SG_F64 ? 8 : 4;
    const int64_t n = A->n_rows;
    if (row_hi < 0) row_hi = n;   // the whole matrix
    PairList pl;
    pl.row_lo = (uint32_t)row_lo;
    pl.row_hi = (uint32_t)row_hi;
    pl.row_step = row_step > 1 ? (uint32_t)row_step : 1u;
    // A name list has a few matches per row, but hubs of identical names have h^2 / 2 pairs each, and the largest hubs
    // grow with the list (5 M synthetic names: 53 M pairs above 0.8 for 18 M matches kept -- with room for 8 n pairs the
    // pass was thrown away after 630 ms and the one-sided form took another 1300; scripts/full_configs.py).  The list
    // costs nothing until it is written: room for 64 n pairs, at most a sixteenth of the device memory.
    // (every row through the exact kernel = thresholds below the pruned kernel's: at 0.3 a name has 70 and more matches
    //  above it on average, 200 k names overran 64 n: room for 512 n there)
    // ... and where a row's own matches travel through the list as well (top_n above one register list), or the threshold is
    //  below the name-matching range (the tile-by-tile form's index): 256 n
    int64_t cap = (exact_all ? 512 : ((keep > SG_TOPN_LANES || Bt->tile_form) ? 256 : 64)) * n + ((int64_t)1 << 20);
    if (ctx->total_mem > 0) {
        const int64_t by_memory = (int64_t)(ctx->total_mem / 16 / (8 + vs));
        if (cap > by_memory) cap = by_memory;
    }
    if (cap < 8 * n + ((int64_t)1 << 20)) cap = 8 * n + ((int64_t)1 << 20);
    bool cap_forced = false;
    if (const char *v = ctx->opt("SG_SYM_PAIR_CAP"))   // test hook: a list that is too small
        if (atoll(v) > 0) {
            cap = atoll(v);
            cap_forced = true;
        }
    if (cap >= ((int64_t)1 << 31)) cap = ((int64_t)1 << 31) - 1;   // list offsets are 32-bit
    // every wave of the kernel holds one open chunk: count those in
    pl.chunks = (uint32_t)(cap / SG_PAIR_CHUNK);
    if (!cap_forced) pl.chunks += 2u * pruned_grid(ctx, Bt->tile_log2, n, Bt->fold_log2, A->dtype) + (uint32_t)ctx->num_cu * 4u +
                                  sg_spgemm_exact_selfjoin_grid(ctx, Bt);   // (every wave of every launch holds one open chunk)
    if (pl.chunks < 1) pl.chunks = 1;
    cap = (int64_t)pl.chunks * SG_PAIR_CHUNK;
    // [0] row counter [1] flagged count [2..3] pairs [4] chunks handed out [5] row counter of the exact kernel's launch;
    // [64, 80) the pair list as a struct; [128 ..) entries per chunk
    uint32_t *words = nullptr;
    uint32_t *cnt = nullptr, *cursor = nullptr;
    int32_t *lcol = nullptr;
    void *lval = nullptr;
    uint32_t *flagged_rows = nullptr;
    int st = sg_alloc(ctx, (size_t)128 + pl.chunks, &words);   // ([64, 80): the SgPairSink handed to the kernels)
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)cap, &pl.d_i);
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)cap, &pl.d_j);
    if (st == SG_OK) st = ctx->alloc((size_t)cap * vs, &pl.d_s);
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)n + 2, &cnt);
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)n + 2, &cursor);
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)n + 2, &flagged_rows);
    unsigned long long *h = (unsigned long long *)ctx->h_stat_words;   // pinned; read before anything else uses it
    auto cleanup = [&]() {
        ctx->release(words);
        ctx->release(pl.d_i);
        ctx->release(pl.d_j);
        ctx->release(pl.d_s);
        ctx->release(cnt);
        ctx->release(cursor);
        ctx->release(flagged_rows);
        ctx->release(lcol);
        ctx->release(lval);
    };
    if (st != SG_OK) {
        cleanup();
        return st;
    }
    pl.d_totals = (unsigned long long *)(words + 2);
    pl.d_chunks_used = words + 4;
    pl.d_row_count = cnt;
    pl.d_chunk_count = words + 128;
    // (a cache line of its own, 256 bytes from the counters: every access to the line of the row counter and the chunk
    //  counter queues behind their atomics -- with the struct next to them the kernel took twice its time)
    pl.d_sink = reinterpret_cast<const SgPairSink *>(words + 64);
    static_assert(sizeof(SgPairSink) <= 64, "the pair list's struct must fit the sixteen words reserved for it");
    unsigned long long *d_stats3 = nullptr;   // this pass's own statistics: they only count when the pass does
    st = sg_alloc(ctx, (size_t)8, &d_stats3);   // [0..2] rows / postings / survivors, [4] pairs scored exactly
    hipError_t e = hipSuccess;
    if (st == SG_OK) {
        st = SG_ZERO4(ctx, words, (128 + (size_t)pl.chunks) * sizeof(uint32_t), d_stats3, 8 * sizeof(unsigned long long), cnt,
                      sizeof(uint32_t) * (size_t)(n + 2), cursor, sizeof(uint32_t) * (size_t)(n + 2));   // (one launch, not four)
    }
    const float s_budget = prune_budget(Bt, threshold, delta);
    SgPairSink sink;
    sink.d_i = pl.d_i;
    sink.d_j = pl.d_j;
    sink.d_s = pl.d_s;
    sink.d_row_count = pl.d_row_count;
    sink.d_chunk_count = pl.d_chunk_count;
    sink.d_chunks_used = pl.d_chunks_used;
    sink.d_totals = pl.d_totals;
    sink.chunks = pl.chunks;
    if (st == SG_OK) {
        hipLaunchKernelGGL(pair_sink_kernel, dim3(1), dim3(1), 0, ctx->stream, sink, (SgPairSink *)(words + 64));
        if (hipGetLastError() != hipSuccess) st = SG_ERR_HIP;
    }
    if (st == SG_OK && exact_all) {
        SgTimer kt(ctx, SG_K_SPGEMM_KERNEL);
        st = sg_postings_ensure_full(ctx, Bt);
        // Large matrices: the most expensive rows -- the last 0.4 % of the positions, which see every column -- go first, in a
        // launch of their own, and what THEY put into the pair list says whether the list will hold the pass: a row at
        // position p mirrors the matches j < p, so the whole pass writes about n / (2 k) times what the last k rows do.  At
        // thresholds far below name matching the pairs outgrow any list (5 M names at 0.38: more than 1.5 G pairs; the pass
        // wrote 18 GB for 6.5 s before it ran out of chunks, and the one-sided form took its 10 s after that): then the
        // form is called off here, for 1 % of its work.  [6] / [8]: the two lists' lengths, [5] / [7]: their row counters.
        uint32_t k_first = 0;
        if (n >= (int64_t)1000000 && !cap_forced) k_first = (uint32_t)(n / 256 > 4096 ? n / 256 : 4096);
        if (const char *v = ctx->opt("SG_EXACT_SYM_PILOT_ROWS")) k_first = (uint32_t)atoll(v) < (uint32_t)n ? (uint32_t)atoll(v) : 0u;   // (test hook)
        auto list_and_launch = [&](uint32_t first, uint32_t count, uint32_t *len_word, uint32_t *counter_word) {
            hipLaunchKernelGGL(all_rows_descending_kernel, dim3((count + 255u) / 256u), dim3(256), 0, ctx->stream, (uint32_t)n, first, count,
                               flagged_rows + first, len_word, words + 1);
            if (hipGetLastError() != hipSuccess) return (int)SG_ERR_HIP;
            return sg_spgemm_exact_selfjoin_rows(ctx, A, Bt, keep, r, threshold, counter_word, flagged_rows + first, len_word, sink,
                                                 /*all_rows=*/true);
        };
        if (st == SG_OK && n > 0 && k_first > 0) {
            st = list_and_launch(0u, k_first, words + 6, words + 5);
            if (st == SG_OK) {
                e = hipMemcpyAsync(h, words, 24, hipMemcpyDeviceToHost, ctx->stream);
                if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
                if (e != hipSuccess) st = SG_ERR_HIP;
            }
            if (st == SG_OK) {
                const double open_chunks = 0.5 * (double)sg_spgemm_exact_selfjoin_grid(ctx, Bt);   // (every wave's last chunk is half full on average)
                const double used = (double)(uint32_t)h[2] > open_chunks ? (double)(uint32_t)h[2] - open_chunks : 0.0;
                if (used * (double)n / (2.0 * (double)k_first) > 0.9 * (double)pl.chunks) {
                    ctx->release(d_stats3);
                    cleanup();
                    return SG_OK;   // *done stays false: the caller runs the one-sided form
                }
                st = list_and_launch(k_first, (uint32_t)n - k_first, words + 8, words + 7);
            }
        } else if (st == SG_OK && n > 0) {
            st = list_and_launch(0u, (uint32_t)n, words + 6, words + 5);
        }
    } else if (st == SG_OK) {
        SgTimer kt(ctx, SG_K_SPGEMM_KERNEL);   // the kernel alone (the launch group's timer also covers the second pass)
        auto pruned_pass = [&](const PairList &range) {
            if (A->dtype == SG_F64)
                return dispatch_pruned<double, true>(ctx, A, Bt, keep, r, (double)threshold, s_budget, words, words + 1, flagged_rows,
                                                     d_stats3, range);
            return dispatch_pruned<float, true>(ctx, A, Bt, keep, r, (float)threshold, s_budget, words, words + 1, flagged_rows, d_stats3,
                                                range);
        };
        // The tile-by-tile form on a large matrix (thresholds below the name-matching range, a million rows and more): the
        // last 0.4 % of the positions first, as the exact kernel's form does above, and their pairs decide whether the list
        // will hold the pass -- 5 M names at 0.42 have more than 1 G pairs above the threshold, the pass wrote them for four
        // seconds before it ran out of chunks and was repeated one-sided (scripts/big_low_thresholds.py).
        uint32_t k_first = 0;
        const bool whole = pl.row_lo == 0 && (int64_t)pl.row_hi == n && pl.row_step == 1 && !export_pairs;
        if (Bt->tile_form && whole && n >= (int64_t)1000000 && !cap_forced) k_first = (uint32_t)(n / 256 > 4096 ? n / 256 : 4096);
        if (const char *v = ctx->opt("SG_SYM_PILOT_ROWS"))    // (test hook, any form of the pruned multiply)
            k_first = whole && atoll(v) > 0 && atoll(v) < n ? (uint32_t)atoll(v) : 0u;
        if (k_first > 0) {
            PairList first = pl, rest = pl;
            first.row_lo = (uint32_t)n - k_first;
            rest.row_hi = (uint32_t)n - k_first;
            st = pruned_pass(first);
            if (st == SG_OK) {
                e = hipMemcpyAsync(h, words, 24, hipMemcpyDeviceToHost, ctx->stream);
                if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
                if (e != hipSuccess) st = SG_ERR_HIP;
            }
            if (st == SG_OK) {
                const double open_chunks = 0.5 * (double)pruned_grid(ctx, Bt->tile_log2, (int64_t)k_first, Bt->fold_log2, A->dtype);
                const double used = (double)(uint32_t)h[2] > open_chunks ? (double)(uint32_t)h[2] - open_chunks : 0.0;
                if (used * (double)n / (2.0 * (double)k_first) > 0.9 * (double)pl.chunks) {
                    ctx->release(d_stats3);
                    cleanup();
                    return SG_OK;   // *done stays false: the caller runs the one-sided form
                }
                if (hipMemsetAsync(words, 0, sizeof(uint32_t), ctx->stream) != hipSuccess) st = SG_ERR_HIP;   // the row counter
            }
            if (st == SG_OK) st = pruned_pass(rest);
        } else {
            st = pruned_pass(pl);
        }
    }
    // h[0] = {row counter, flagged}, h[1] = pairs, h[2] = chunks handed out
    auto read_back = [&]() {
        e = hipMemcpyAsync(h, words, 24, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) {
            sg_set_error("symmetric multiply: %s", hipGetErrorString(e));
            st = SG_ERR_HIP;
        }
    };
    if (st == SG_OK) read_back();
    if (st == SG_OK && (uint32_t)(h[0] >> 32) > 0 && !exact_all) {
        // the rows neither launch of the pruned kernel could take (more than 128 non-zeros, more than 64 prefix terms, no
        // room for the fixed-point filter, or -- stream form -- none at all): through the exact kernel, in the same form --
        // pairs (i, j <= i), mirrored ones into the pair list.  Its postings are written now if the index build left them
        // out (name lists have no such rows), and the counts are read again.
        st = sg_postings_ensure_full(ctx, Bt);
        // (a list of long strings hands over most of its rows: then the launch is sized like one over all rows, not like the
        //  usual handful -- 50 k strings of ~170 characters: scripts/family_sweep.py, "very long")
        const bool many = (uint32_t)(h[0] >> 32) > 2u * sg_spgemm_exact_selfjoin_grid(ctx);
        if (st == SG_OK) st = sg_spgemm_exact_selfjoin_rows(ctx, A, Bt, keep, r, threshold, words + 5, flagged_rows, words + 1, sink, many);
        if (st == SG_OK) read_back();
    }
    if (st != SG_OK) {
        ctx->release(d_stats3);
        cleanup();
        return st;
    }
    const unsigned long long n_pairs = h[1];
    const uint32_t chunks_used = (uint32_t)h[2];
    if (chunks_used > pl.chunks) {   // more pairs than the list holds (hubs of thousands of identical names)
        ctx->release(d_stats3);
        cleanup();
        return SG_OK;   // *done stays false
    }
    const dim3 pgrid(chunks_used > 0 ? chunks_used : 1);
    if (export_pairs) {
        // ---- multi-GPU: hand the pair list out (the caller gathers the lists of all ranks, then sg_selfjoin_merge)
        const int W = A->dtype == SG_F64 ? 4 : 3;
        uint32_t *chunk_start = nullptr;
        int32_t *flat = nullptr;
        st = sg_alloc(ctx, (size_t)chunks_used + 2, &chunk_start);
        if (st == SG_OK && chunks_used > 0)
            st = sg_exclusive_scan_u32(ctx, pl.d_chunk_count, chunk_start, chunks_used, nullptr);
        if (st == SG_OK) st = sg_alloc(ctx, (size_t)(n_pairs + 1) * W, &flat);
        if (st == SG_OK && chunks_used > 0) {
            if (A->dtype == SG_F64)
                hipLaunchKernelGGL(pairs_export_kernel<double>, pgrid, dim3(SG_PAIR_CHUNK), 0, ctx->stream, pl.d_i, pl.d_j,
                                   (const double *)pl.d_s, pl.d_chunk_count, chunk_start, flat,
                                   (const unsigned long long *)d_stats3, (const uint32_t *)(words + 1), stats);
            else
                hipLaunchKernelGGL(pairs_export_kernel<float>, pgrid, dim3(SG_PAIR_CHUNK), 0, ctx->stream, pl.d_i, pl.d_j,
                                   (const float *)pl.d_s, pl.d_chunk_count, chunk_start, flat,
                                   (const unsigned long long *)d_stats3, (const uint32_t *)(words + 1), stats);
            if (hipGetLastError() != hipSuccess) st = SG_ERR_HIP;
        } else if (st == SG_OK && (hipMemcpyAsync(stats, d_stats3, 3 * sizeof(unsigned long long), hipMemcpyDeviceToDevice,
                                                  ctx->stream) != hipSuccess ||
                                   hipMemcpyAsync(stats + 4, d_stats3 + 4, sizeof(unsigned long long), hipMemcpyDeviceToDevice,
                                                  ctx->stream) != hipSuccess ||
                                   hipMemcpyAsync(stats + 3, words + 1, 4, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess))
            st = SG_ERR_HIP;   // (no pairs, no export kernel: the statistics go by copy)
        ctx->release(chunk_start);
        ctx->release(d_stats3);
        cleanup();
        if (st != SG_OK) {
            ctx->release(flat);
            return st;
        }
        *export_pairs = flat;
        *export_n = (int64_t)n_pairs;
        *done = true;
        return SG_OK;
    }
    // ---- pass 2
    st = sg_exclusive_scan_u32(ctx, cnt, cnt, n + 1, nullptr);   // cnt becomes ptr (n + 1 entries)
    const size_t n_list = (size_t)(n_pairs + 64);
    if (st == SG_OK) st = sg_alloc(ctx, n_list, &lcol);
    if (st == SG_OK) st = ctx->alloc(n_list * vs, &lval);
    if (st == SG_OK) {
        unsigned sgrid = (unsigned)((n + 63) / 64 > 0 ? (n + 63) / 64 : 1);
        if (A->dtype == SG_F64) {
            hipLaunchKernelGGL(pairs_fill_kernel<double>, pgrid, dim3(SG_PAIR_CHUNK), 0, ctx->stream, pl.d_i, pl.d_j,
                               (const double *)pl.d_s, pl.d_chunk_count, cnt, cursor, lcol, (double *)lval);
            hipLaunchKernelGGL(pairs_select_kernel<double>, dim3(sgrid), dim3(64), 0, ctx->stream, cnt, lcol, (const double *)lval,
                               (uint32_t)n, keep, r->stride, r->d_cols, (double *)r->d_vals, r->d_counts,
                               (const unsigned long long *)d_stats3, (const uint32_t *)(words + 1), stats);
        } else {
            hipLaunchKernelGGL(pairs_fill_kernel<float>, pgrid, dim3(SG_PAIR_CHUNK), 0, ctx->stream, pl.d_i, pl.d_j,
                               (const float *)pl.d_s, pl.d_chunk_count, cnt, cursor, lcol, (float *)lval);
            hipLaunchKernelGGL(pairs_select_kernel<float>, dim3(sgrid), dim3(64), 0, ctx->stream, cnt, lcol, (const float *)lval,
                               (uint32_t)n, keep, r->stride, r->d_cols, (float *)r->d_vals, r->d_counts,
                               (const unsigned long long *)d_stats3, (const uint32_t *)(words + 1), stats);
        }
        if (hipGetLastError() != hipSuccess) st = SG_ERR_HIP;   // (pairs_select also publishes the pass's statistics)
    }
    ctx->release(d_stats3);
    cleanup();
    if (st == SG_OK) *done = true;
    return st;
}


// Second pass of the multi-GPU self-join: merge the mirrored pairs (of all ranks) whose row lies in [row_lo, row_hi)
// into those rows of `r` (which hold their own matches from sg_spgemm_pruned_symmetric over the same range).
int sg_selfjoin_merge_pairs(sg_ctx *ctx, sg_topn *r, const int32_t *d_pairs, int64_t n_pairs, int64_t row_lo, int64_t row_hi,
                            const uint32_t *pos_of, int64_t row_step) {
    const uint32_t step = row_step > 1 ? (uint32_t)row_step : 1u;
    const int64_t n = r->n_rows;
    if (n_pairs <= 0 || row_hi <= row_lo) return SG_OK;
    const size_t vs = r->dtype == SG_F64 ? 8 : 4;
    uint32_t *cnt = nullptr, *cursor = nullptr;
    int32_t *lcol = nullptr;
    void *lval = nullptr;
    int st = sg_alloc(ctx, (size_t)n + 2, &cnt);
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)n + 2, &cursor);
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)n_pairs + 64, &lcol);
    if (st == SG_OK) st = ctx->alloc(((size_t)n_pairs + 64) * vs, &lval);
    if (st == SG_OK) {
        st = SG_ZERO2(ctx, cnt, sizeof(uint32_t) * (size_t)(n + 2), cursor, sizeof(uint32_t) * (size_t)(n + 2));
    }
    if (st == SG_OK) {
        const unsigned pg = (unsigned)((n_pairs + 255) / 256);
        if (r->dtype == SG_F64)
            hipLaunchKernelGGL(pairs_flat_count_kernel<4>, dim3(pg), dim3(256), 0, ctx->stream, d_pairs, n_pairs, (uint32_t)row_lo,
                               (uint32_t)row_hi, step, pos_of, cnt);
        else
            hipLaunchKernelGGL(pairs_flat_count_kernel<3>, dim3(pg), dim3(256), 0, ctx->stream, d_pairs, n_pairs, (uint32_t)row_lo,
                               (uint32_t)row_hi, step, pos_of, cnt);
        st = sg_exclusive_scan_u32(ctx, cnt, cnt, n + 1, nullptr);
        if (st == SG_OK) {
            const unsigned sgrid = (unsigned)((n + 63) / 64 > 0 ? (n + 63) / 64 : 1);
            if (r->dtype == SG_F64) {
                hipLaunchKernelGGL(pairs_flat_fill_kernel<double>, dim3(pg), dim3(256), 0, ctx->stream, d_pairs, n_pairs,
                                   (uint32_t)row_lo, (uint32_t)row_hi, step, pos_of, cnt, cursor, lcol, (double *)lval);
                hipLaunchKernelGGL(pairs_select_kernel<double>, dim3(sgrid), dim3(64), 0, ctx->stream, cnt, lcol,
                                   (const double *)lval, (uint32_t)n, r->stride, r->stride, r->d_cols, (double *)r->d_vals,
                                   r->d_counts, (const unsigned long long *)nullptr, (const uint32_t *)nullptr,
                                   (unsigned long long *)nullptr);
            } else {
                hipLaunchKernelGGL(pairs_flat_fill_kernel<float>, dim3(pg), dim3(256), 0, ctx->stream, d_pairs, n_pairs,
                                   (uint32_t)row_lo, (uint32_t)row_hi, step, pos_of, cnt, cursor, lcol, (float *)lval);
                hipLaunchKernelGGL(pairs_select_kernel<float>, dim3(sgrid), dim3(64), 0, ctx->stream, cnt, lcol,
                                   (const float *)lval, (uint32_t)n, r->stride, r->stride, r->d_cols, (float *)r->d_vals,
                                   r->d_counts, (const unsigned long long *)nullptr, (const uint32_t *)nullptr,
                                   (unsigned long long *)nullptr);
            }
            if (hipGetLastError() != hipSuccess) st = SG_ERR_HIP;
        }
    }
    ctx->release(cnt);
    ctx->release(cursor);
    ctx->release(lcol);
    ctx->release(lval);
    return st;
}
