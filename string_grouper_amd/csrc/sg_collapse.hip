// Exact collapse of identical right-hand rows (round 3).
//
// Identical strings have identical TF-IDF rows and therefore identical scores against everything.  The reference's
// showcase data is full of them (README.md:80-95 of the reference: groups of 1 747 / 916 / 652 identical names), and a hub
// of h identical names makes the multiply score h^2 / 2 pairs that all carry the same number -- a tenth of a 663 k list
// being one name took 391 ms (shape sweep) against 15 ms without it.  So the index (sg_postings.hip) is built
// over ONE representative per group of identical rows, the multiply runs on groups, and its result is expanded:
//
//   * grouping: a 64-bit hash per row (sixteen lanes per row), a stable radix sort of (hash, row) (rocPRIM, like the
//     wide-key vocabulary), a head flag where a sorted row differs from its predecessor -- compared entry by entry, the hash
//     only brings candidates together; rows that collide without being equal stay separate groups -- and a scan;
//     the representative of a group is its lowest row, groups are numbered by ascending representative, members listed
//     ascending;
//   * multiply: A x U^T (one-sided) or U x U^T (self-join; all forms of the pruned multiply apply) with the caller's top_n:
//     a row of the result over groups, ordered (score descending, group ascending), holds every group that can contribute
//     to the row's top_n columns -- each group expands to at least one column, and the representative of a group among the
//     best top_n columns is itself among them;
//   * expansion: per result row the groups' members in (score descending, column ascending) order -- groups of equal
//     score are merged by column -- cut at top_n: exactly what the multiply over all rows returns (the canonical order
//     this build defines, oracle/oracle.py); in the self-join every member of a group gets its representative's row.
//
// Off when fewer than 3 % of the rows are repeats (the grouping costs ~0.3 ms at 663 k, the multiply grows with the
// square of the rows): SG_COLLAPSE=0 / 1 force it.
#include "sg_internal.h"
#include "sg_scan.h"

template <typename T>
__global__ void __launch_bounds__(256) row_hash_kernel(const int64_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                                                       const T *__restrict__ data, int64_t n_rows, uint64_t *__restrict__ hash,
                                                       uint32_t *__restrict__ row_id /* null: not wanted */) {
    const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const int sub = threadIdx.x & 15;
    uint64_t h = 0;
    if (r < n_rows) {
        const int64_t lo = indptr[r], hi = indptr[r + 1];
        for (int64_t p = lo + sub; p < hi; p += 16) {
            uint64_t v;
            if (sizeof(T) == 4) v = (uint64_t)__float_as_uint((float)data[p]);
            else v = (uint64_t)__double_as_longlong((double)data[p]);
            uint64_t x = ((uint64_t)(uint32_t)indices[p] << 32) ^ v ^ ((uint64_t)(p - lo) * 0x9E3779B97F4A7C15ull);
            x ^= x >> 33;
            x *= 0xff51afd7ed558ccdull;
            x ^= x >> 33;
            x *= 0xc4ceb9fe1a85ec53ull;
            x ^= x >> 33;
            h += x;   // (the position is mixed in: the sum over the lanes is order-free but not content-free)
        }
    }
#pragma unroll
    for (int d = 8; d > 0; d >>= 1) {
        const uint32_t lo32 = (uint32_t)__shfl_xor((int)(uint32_t)h, d, 64), hi32 = (uint32_t)__shfl_xor((int)(uint32_t)(h >> 32), d, 64);
        h += ((uint64_t)hi32 << 32) | lo32;
    }
    if (r < n_rows && sub == 0) {
        const int64_t len = indptr[r + 1] - indptr[r];
        hash[r] = h ^ ((uint64_t)len * 0xD6E8FEB86659FD93ull);
        if (row_id) row_id[r] = (uint32_t)r;
    }
}

// ---- grouping through a hash table (round 4; the sort-based path below stays for lists with very large groups).
// The stable sort of (hash, row) was 21 launches and 0.17 ms of the 0.42 ms the grouping cost at 663 k.  An open-addressing
// table keyed by the 64-bit row hash does the same job: a slot belongs to ONE hash value (that of the row it names) and
// ends up naming the LOWEST row with that hash (atomicMin) -- the group's representative, as before; every row then
// checks itself against that row entry by entry (a row whose hash collides with different content becomes a group of its
// own: nothing is ever merged on the hash alone).  Groups are numbered by ascending representative and list their
// members ascending as before: members are scattered in arrival order and every group of several members is sorted
// (two members: a swap; up to 32: by its thread; up to 8192: by a workgroup in LDS; a larger one -- a hub -- by a
// flag / prefix-sum / scatter pass of its own; a list with dozens of such hubs takes the sort-based path).
#define SG_GROUP_SORT_LDS 8192
#define SG_GROUP_LARGE_MAX 28u      // very large groups listed per call (more: the sort-based path)
#define SG_GROUP_NO_SLOT 0xFFFFFFFFu

// Round 6: hash, insert and verify in ONE pass over the rows (rounds 4-5: row_hash_kernel, group_insert_kernel,
// group_verify_kernel -- the rows read twice, 0.15 ms of the 0.3 ms the grouping cost at 663 k, 1.1 of 1.8 ms at 5 M).
// A slot is 64 bits, {tag = upper half of the row's hash, ~row}, 0 = empty: the tag tells whose slot it is without a
// hash array to look the holder's hash up in, and with ~row in the low half an atomic MAX over rows of one tag keeps the
// LOWEST row.  Sixteen lanes per row: they hash the row, then walk the table together -- an empty slot is claimed
// (compare-and-swap), a slot of another tag is passed, a slot of the row's own tag names a row whose CONTENT is compared
// entry by entry on the spot: equal -> the row joins the slot (and lowers it if it is the lower row), different -> a
// collision of 53 bits of hash: the row becomes a group of its own, as before.  Every row that ever held or joined a slot
// has been compared equal to a row that held it before, so all of them are equal and the slot's final holder -- the
// lowest -- is their representative: nothing is merged on the hash alone.  slot_of_row[r] = the slot, or SG_GROUP_NO_SLOT
// for a row that stands alone.
// The walk is a loop the WAVE leaves together, every group's state in registers (hipcc 7.2 and
// per-lane loops whose result is read behind them).
// TWO rows per sixteen lanes, every stage written for both before the first result is used: a row is a chain of dependent
// round trips (row pointers, entries, the slot, the holder's row pointers, its entries) and with one row per sixteen lanes
// the kernel ran at the latency of that chain (0.87 ms for 0.76 GB at 5 M).
#define SG_GROUP_ROWS 2
template <typename T>
__global__ void __launch_bounds__(256) group_rows_kernel(const int64_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                                                         const T *__restrict__ data, int64_t n_rows, unsigned long long *table,
                                                         uint32_t mask, uint32_t *__restrict__ slot_of_row) {
    const int64_t grp = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const int sub = threadIdx.x & 15;
    const int lead = (int)(threadIdx.x & 48u);          // the group's first lane inside its wave
    int64_t r[SG_GROUP_ROWS], lo[SG_GROUP_ROWS];
    int32_t len[SG_GROUP_ROWS];
    bool valid[SG_GROUP_ROWS];
#pragma unroll
    for (int i = 0; i < SG_GROUP_ROWS; ++i) {
        r[i] = grp * SG_GROUP_ROWS + i;
        valid[i] = r[i] < n_rows;
        lo[i] = valid[i] ? indptr[r[i]] : 0;
        len[i] = valid[i] ? (int32_t)(indptr[r[i] + 1] - lo[i]) : 0;
    }
    // a row's first two rounds of entries stay in registers for the comparisons (rows of up to 32 entries: all of it)
    int32_t k0[SG_GROUP_ROWS], k1[SG_GROUP_ROWS];
    T v0[SG_GROUP_ROWS], v1[SG_GROUP_ROWS];
#pragma unroll
    for (int i = 0; i < SG_GROUP_ROWS; ++i) {
        k0[i] = k1[i] = -1;
        v0[i] = v1[i] = (T)0;
        if (sub < len[i]) {
            k0[i] = indices[lo[i] + sub];
            v0[i] = data[lo[i] + sub];
        }
        if (sub + 16 < len[i]) {
            k1[i] = indices[lo[i] + sub + 16];
            v1[i] = data[lo[i] + sub + 16];
        }
    }
    unsigned long long mine[SG_GROUP_ROWS];
    uint32_t s[SG_GROUP_ROWS], result[SG_GROUP_ROWS];
    bool walking[SG_GROUP_ROWS];
#pragma unroll
    for (int i = 0; i < SG_GROUP_ROWS; ++i) {
        uint64_t h = 0;
        for (int32_t e = sub; e < len[i]; e += 16) {
            const int32_t k = e < 16 ? k0[i] : (e < 32 ? k1[i] : indices[lo[i] + e]);
            const T v = e < 16 ? v0[i] : (e < 32 ? v1[i] : data[lo[i] + e]);
            uint64_t vb;
            if (sizeof(T) == 4) vb = (uint64_t)__float_as_uint((float)v);
            else vb = (uint64_t)__double_as_longlong((double)v);
            uint64_t x = ((uint64_t)(uint32_t)k << 32) ^ vb ^ ((uint64_t)e * 0x9E3779B97F4A7C15ull);
            x ^= x >> 33;
            x *= 0xff51afd7ed558ccdull;
            x ^= x >> 33;
            x *= 0xc4ceb9fe1a85ec53ull;
            x ^= x >> 33;
            h += x;   // (the position is mixed in: the sum over the lanes is order-free but not content-free)
        }
#pragma unroll
        for (int d = 8; d > 0; d >>= 1) {
            const uint32_t lo32 = (uint32_t)__shfl_xor((int)(uint32_t)h, d, 64), hi32 = (uint32_t)__shfl_xor((int)(uint32_t)(h >> 32), d, 64);
            h += ((uint64_t)hi32 << 32) | lo32;
        }
        h ^= (uint64_t)(uint32_t)len[i] * 0xD6E8FEB86659FD93ull;
        h ^= h >> 31;
        mine[i] = ((unsigned long long)(uint32_t)(h >> 32) << 32) | (unsigned long long)(~(uint32_t)r[i]);
        s[i] = (uint32_t)h & mask;
        result[i] = SG_GROUP_NO_SLOT;
        walking[i] = valid[i];
    }
    while (__ballot(walking[0] || walking[1]) != 0) {
        // what the slots hold (claimed on the spot when empty): both rows' round trips together
        unsigned long long cur[SG_GROUP_ROWS];
#pragma unroll
        for (int i = 0; i < SG_GROUP_ROWS; ++i) {
            cur[i] = 0;
            if (walking[i] && sub == 0) cur[i] = __hip_atomic_load(&table[s[i]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int i = 0; i < SG_GROUP_ROWS; ++i)
            if (walking[i] && sub == 0 && cur[i] == 0ull) cur[i] = atomicCAS(&table[s[i]], 0ull, mine[i]);   // (0: the slot is this row's now)
        uint32_t c_lo[SG_GROUP_ROWS], c_hi[SG_GROUP_ROWS];
        bool cmp[SG_GROUP_ROWS];
        int64_t o[SG_GROUP_ROWS], olo[SG_GROUP_ROWS];
        int32_t olen[SG_GROUP_ROWS];
#pragma unroll
        for (int i = 0; i < SG_GROUP_ROWS; ++i) {
            c_lo[i] = (uint32_t)__shfl((int)(uint32_t)cur[i], lead, 64);
            c_hi[i] = (uint32_t)__shfl((int)(uint32_t)(cur[i] >> 32), lead, 64);
            // the holder has this row's tag: the same content?
            cmp[i] = walking[i] && !(c_lo[i] == 0u && c_hi[i] == 0u) && c_hi[i] == (uint32_t)(mine[i] >> 32);
            o[i] = (int64_t)(~c_lo[i]);
            olo[i] = cmp[i] ? indptr[o[i]] : 0;
            olen[i] = cmp[i] ? (int32_t)(indptr[o[i] + 1] - olo[i]) : -1;
        }
        int32_t ok0[SG_GROUP_ROWS], ok1[SG_GROUP_ROWS];
        T ov0[SG_GROUP_ROWS], ov1[SG_GROUP_ROWS];
#pragma unroll
        for (int i = 0; i < SG_GROUP_ROWS; ++i) {
            const bool go = cmp[i] && olen[i] == len[i];
            ok0[i] = ok1[i] = -1;
            ov0[i] = ov1[i] = (T)0;
            if (go && sub < len[i]) {
                ok0[i] = indices[olo[i] + sub];
                ov0[i] = data[olo[i] + sub];
            }
            if (go && sub + 16 < len[i]) {
                ok1[i] = indices[olo[i] + sub + 16];
                ov1[i] = data[olo[i] + sub + 16];
            }
        }
#pragma unroll
        for (int i = 0; i < SG_GROUP_ROWS; ++i) {
            if (!walking[i]) continue;
            if (c_lo[i] == 0u && c_hi[i] == 0u) {
                result[i] = s[i];
                walking[i] = false;
            } else if (cmp[i]) {
                bool same = olen[i] == len[i];
                if (same) {
                    same = ok0[i] == k0[i] && ov0[i] == v0[i] && ok1[i] == k1[i] && ov1[i] == v1[i];
                    for (int32_t e = sub + 32; e < len[i]; e += 16)
                        same = same && indices[olo[i] + e] == indices[lo[i] + e] && data[olo[i] + e] == data[lo[i] + e];
                }
#pragma unroll
                for (int d = 8; d > 0; d >>= 1) same = same && (__shfl_xor((int)same, d, 64) != 0);
                if (same) {
                    if (sub == 0 && o[i] > r[i]) atomicMax(&table[s[i]], mine[i]);   // (~row: the max keeps the lowest row)
                    result[i] = s[i];
                }
                walking[i] = false;     // (different content under one tag: the row stands alone)
            } else {
                s[i] = (s[i] + 1u) & mask;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < SG_GROUP_ROWS; ++i)
        if (valid[i] && sub == 0) slot_of_row[r[i]] = result[i];
}

// the representative of row r once every row has been through group_rows_kernel
__device__ __forceinline__ uint32_t sg_group_rep(const unsigned long long *__restrict__ table, const uint32_t *__restrict__ slot_of_row,
                                                 int64_t r) {
    const uint32_t s = slot_of_row[r];
    return s == SG_GROUP_NO_SLOT ? (uint32_t)r : ~(uint32_t)table[s];
}
// the scan over "row r is a representative" computes its input itself and leaves rep_of_row behind
struct GroupRepLoad {
    const unsigned long long *table;
    const uint32_t *slot_of_row;
    uint32_t *rep_of_row;
    __device__ __forceinline__ uint32_t operator()(int64_t r) const {
        const uint32_t rep = sg_group_rep(table, slot_of_row, r);
        rep_of_row[r] = rep;
        return rep == (uint32_t)r ? 1u : 0u;
    }
};

// members in arrival order (a group of one needs no ticket)
__global__ void __launch_bounds__(256) group_scatter_kernel(const uint32_t *__restrict__ gid, const uint32_t *__restrict__ group_ptr,
                                                            int64_t n_rows, uint32_t *cursor, uint32_t *__restrict__ members) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    const uint32_t g = gid[r];
    const uint32_t lo = group_ptr[g], m = group_ptr[g + 1] - lo;
    if (m > (uint32_t)SG_GROUP_SORT_LDS) return;        // a very large group: listed by a compaction of its own (large_group_*)
    members[lo + (m == 1u ? 0u : atomicAdd(&cursor[g], 1u))] = (uint32_t)r;
}

// A group too large for the workgroup sort (a hub of thousands of identical names): its members in ascending order are the
// rows r with gid[r] == g in row order -- a flag per row, a prefix sum, a scatter (tickets on one counter would be tens of
// thousands of returning atomics on one address, and the list would still have to be sorted).
__global__ void __launch_bounds__(256) large_group_flag_kernel(const uint32_t *__restrict__ gid, int64_t n_rows, uint32_t g,
                                                               uint32_t *__restrict__ flag) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n_rows) flag[r] = gid[r] == g ? 1u : 0u;
}
__global__ void __launch_bounds__(256) large_group_fill_kernel(const uint32_t *__restrict__ gid, const uint32_t *__restrict__ at,
                                                               int64_t n_rows, uint32_t g, const uint32_t *__restrict__ group_ptr,
                                                               uint32_t *__restrict__ members) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n_rows && gid[r] == g) members[group_ptr[g] + at[r]] = (uint32_t)r;
}

// a thread per group: members ascending.  Groups of more than 32 are queued for the workgroup sort; words[0] = queue
// length, words[1] = largest group.
__global__ void __launch_bounds__(256) group_sort_small_kernel(const uint32_t *__restrict__ group_ptr,
                                                               const uint32_t *__restrict__ n_u_dev /* the number of groups: known to the device, not yet to the host */,
                                                               uint32_t *members, uint32_t *words, uint32_t *__restrict__ queue) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (int64_t)*n_u_dev) return;
    const uint32_t lo = group_ptr[g], m = group_ptr[g + 1] - lo;
    if (m < 2u) return;
    if (m == 2u) {
        const uint32_t a = members[lo], b = members[lo + 1];
        if (a > b) {
            members[lo] = b;
            members[lo + 1] = a;
        }
        return;
    }
    if (m > 32u) {
        atomicMax(&words[1], m);
        if (m <= (uint32_t)SG_GROUP_SORT_LDS) queue[atomicAdd(&words[0], 1u)] = (uint32_t)g;
        else {                                               // words[2] = how many, words[3 ..] the first of them
            const uint32_t q = atomicAdd(&words[2], 1u);
            if (q < SG_GROUP_LARGE_MAX) words[3 + q] = (uint32_t)g;
        }
        return;
    }
    for (uint32_t i = 1; i < m; ++i) {           // insertion sort in place (the segment is this thread's alone)
        const uint32_t v = members[lo + i];
        uint32_t j = i;
        while (j > 0 && members[lo + j - 1] > v) {
            members[lo + j] = members[lo + j - 1];
            --j;
        }
        members[lo + j] = v;
    }
}

// a workgroup per queued group: bitonic sort in LDS
__global__ void __launch_bounds__(256) group_sort_lds_kernel(const uint32_t *__restrict__ group_ptr, uint32_t *members,
                                                             const uint32_t *__restrict__ words, const uint32_t *__restrict__ queue) {
    __shared__ uint32_t buf[SG_GROUP_SORT_LDS];
    const uint32_t n_q = words[0];
    for (uint32_t q = blockIdx.x; q < n_q; q += gridDim.x) {
        const uint32_t g = queue[q];
        const uint32_t lo = group_ptr[g], m = group_ptr[g + 1] - lo;
        uint32_t P = 64;
        while (P < m) P <<= 1;
        for (uint32_t i = threadIdx.x; i < P; i += blockDim.x) buf[i] = i < m ? members[lo + i] : 0xFFFFFFFFu;
        __syncthreads();
        for (uint32_t k = 2; k <= P; k <<= 1)
            for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                for (uint32_t i = threadIdx.x; i < P; i += blockDim.x) {
                    const uint32_t x = i ^ j;
                    if (x > i) {
                        const uint32_t a = buf[i], b = buf[x];
                        const bool up = (i & k) == 0;
                        if ((a > b) == up) {
                            buf[i] = b;
                            buf[x] = a;
                        }
                    }
                }
                __syncthreads();
            }
        for (uint32_t i = threadIdx.x; i < m; i += blockDim.x) members[lo + i] = buf[i];
        __syncthreads();
    }
}

// head[s] = 1 when the row at sorted position s starts a group: first of all, another hash, or another content
template <typename T>
__global__ void __launch_bounds__(256) group_heads_kernel(const int64_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                                                          const T *__restrict__ data, int64_t n_rows,
                                                          const uint64_t *__restrict__ hash_sorted,
                                                          const uint32_t *__restrict__ row_sorted, uint32_t *__restrict__ head) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_rows) return;
    uint32_t is_head = 1;
    if (s > 0 && hash_sorted[s] == hash_sorted[s - 1]) {
        const int64_t a = row_sorted[s], b = row_sorted[s - 1];
        const int64_t la = indptr[a], lb = indptr[b];
        const int64_t n = indptr[a + 1] - la;
        bool same = n == indptr[b + 1] - lb;
        for (int64_t e = 0; same && e < n; ++e) same = indices[la + e] == indices[lb + e] && data[la + e] == data[lb + e];
        is_head = same ? 0u : 1u;
    }
    head[s] = is_head;
}

// run[s] = inclusive scan of head - 1.  head_pos[run] = position of the run's head.
__global__ void __launch_bounds__(256) head_pos_kernel(const uint32_t *__restrict__ head, const uint32_t *__restrict__ run_excl,
                                                       int64_t n_rows, uint32_t *__restrict__ head_pos) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n_rows && head[s]) head_pos[run_excl[s]] = (uint32_t)s;   // run id of a head = heads before it
}

// per row: its representative, its rank inside the group; is_rep[rep] = 1
__global__ void __launch_bounds__(256) group_members_kernel(const uint32_t *__restrict__ head, const uint32_t *__restrict__ run_excl,
                                                            const uint32_t *__restrict__ head_pos,
                                                            const uint32_t *__restrict__ row_sorted, int64_t n_rows,
                                                            uint32_t *__restrict__ rep_of_row, uint32_t *__restrict__ rank_of_row,
                                                            uint32_t *__restrict__ is_rep) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_rows) return;
    const uint32_t run = run_excl[s] + head[s] - 1u;     // inclusive scan - 1
    const uint32_t hp = head_pos[run];
    const uint32_t row = row_sorted[s], rep = row_sorted[hp];
    rep_of_row[row] = rep;
    rank_of_row[row] = (uint32_t)s - hp;                  // the sort is stable: rows of a group ascend with s
    if (head[s]) is_rep[row] = 1u;
}

// gid[row] = number of representatives below the row's representative; sizes counted; rep_rows[gid] = representative.
// Table path (indptr given): a representative also leaves where its row starts in the source and how long it is -- what
// the index build's one read of the rows (sg_postings.hip, gather_rows_kernel) and the lazy matrix of the representatives
// need --, and the lengths are summed into the 32 counters of nnz_total (one atomic per wave).
__global__ void __launch_bounds__(256) group_ids_kernel(const uint32_t *__restrict__ rep_of_row, const uint32_t *__restrict__ rep_excl,
                                                        const uint32_t *__restrict__ is_rep /* null: rep_of_row[r] == r says so */,
                                                        int64_t n_rows, uint32_t *__restrict__ gid, uint32_t *__restrict__ size,
                                                        uint32_t *__restrict__ rep_rows, const int64_t *__restrict__ indptr,
                                                        int64_t *__restrict__ rep_start, int32_t *__restrict__ rep_len,
                                                        unsigned long long *nnz_total) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = r < n_rows;
    uint32_t g = 0xFFFFFFFFu;
    bool rep = false;
    uint32_t my_len = 0;
    if (valid) {
        const uint32_t ro = rep_of_row[r];
        g = rep_excl[ro];
        gid[r] = g;
        rep = is_rep ? is_rep[r] != 0u : ro == (uint32_t)r;
        if (rep) {
            rep_rows[g] = (uint32_t)r;
            atomicAdd(&size[g], 1u);          // (one representative per group: an address of its own)
            if (indptr) {
                const int64_t lo = indptr[r];
                my_len = (uint32_t)(indptr[r + 1] - lo);
                rep_start[g] = lo;
                rep_len[g] = (int32_t)my_len;
            }
        }
    }
    if (nnz_total) {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) my_len += (uint32_t)__shfl_xor((int)my_len, d, 64);
        // (32 counters, one per workgroup mod 32: ten thousand waves on ONE word are served one at a time, ~12 ns each --
        //  0.12 ms at 663 k; and every counter on a 128-byte line of its own: atomics on ONE LINE queue up just the same)
        if ((threadIdx.x & 63) == 0 && my_len) atomicAdd(nnz_total + 16u * (blockIdx.x & 31u), (unsigned long long)my_len);
    }
    // the other members: the lanes of a wave that belong to one group send ONE atomic between them -- a hub of 66 000
    // identical names was 66 000 atomics on one word, which are served one at a time (~12 ns each: 0.8 ms of a 2 ms build)
    uint64_t todo = __ballot(valid && !rep);
    while (todo) {
        const int lead = __builtin_ctzll(todo);
        const uint32_t lg = (uint32_t)__builtin_amdgcn_readlane((int)g, lead);
        const uint64_t same = __ballot(valid && !rep && g == lg) & todo;
        if ((int)(threadIdx.x & 63) == lead) atomicAdd(&size[lg], (uint32_t)__popcll(same));
        todo &= ~same;
    }
}

__global__ void __launch_bounds__(256) group_fill_kernel(const uint32_t *__restrict__ gid, const uint32_t *__restrict__ rank_of_row,
                                                         const uint32_t *__restrict__ group_ptr, int64_t n_rows,
                                                         uint32_t *__restrict__ members) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n_rows) members[group_ptr[gid[r]] + rank_of_row[r]] = (uint32_t)r;
}

__global__ void __launch_bounds__(256) unique_len_kernel(const int64_t *__restrict__ indptr, const uint32_t *__restrict__ rep_rows,
                                                         int64_t n_u, int32_t *__restrict__ len) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g < n_u) len[g] = (int32_t)(indptr[rep_rows[g] + 1] - indptr[rep_rows[g]]);
}

template <typename T>
__global__ void __launch_bounds__(256) unique_rows_kernel(const int64_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                                                          const T *__restrict__ data, const uint32_t *__restrict__ rep_rows,
                                                          int64_t n_u, const int64_t *__restrict__ out_ptr,
                                                          int32_t *__restrict__ out_indices, T *__restrict__ out_data) {
    const int64_t g = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const int sub = threadIdx.x & 15;
    if (g >= n_u) return;
    const int64_t src = indptr[rep_rows[g]], n = indptr[rep_rows[g] + 1] - src, dst = out_ptr[g];
    for (int64_t e = sub; e < n; e += 16) {
        out_indices[dst + e] = indices[src + e];
        out_data[dst + e] = data[src + e];
    }
}

int sg_collapse_materialize(sg_ctx *ctx, SgCollapse *c) {
    if (!c || !c->pending_src) return SG_OK;
    const sg_csr *B = c->pending_src;
    sg_csr *m = c->unique;
    // (table path, round 6: the row pointers are pending too -- nothing on the way to the self-join's index reads them)
    if (c->d_rep_len) SG_TRY(sg_exclusive_scan_i32_to_i64(ctx, c->d_rep_len, (int64_t *)m->d_indptr, c->n_u));
    const unsigned gu = (unsigned)((c->n_u * 16 + 255) / 256);
    if (B->dtype == SG_F64)
        hipLaunchKernelGGL(unique_rows_kernel<double>, dim3(gu), dim3(256), 0, ctx->stream, B->d_indptr, B->d_indices,
                           (const double *)B->d_data, (const uint32_t *)c->d_rep_rows, c->n_u, m->d_indptr, (int32_t *)m->d_indices,
                           (double *)m->d_data);
    else
        hipLaunchKernelGGL(unique_rows_kernel<float>, dim3(gu), dim3(256), 0, ctx->stream, B->d_indptr, B->d_indices,
                           (const float *)B->d_data, (const uint32_t *)c->d_rep_rows, c->n_u, m->d_indptr, (int32_t *)m->d_indices,
                           (float *)m->d_data);
    SG_HIP_TRY(hipGetLastError());
    c->pending_src = nullptr;
    return SG_OK;
}

// The rows of a matrix that may be the representatives' matrix of some groups, not written yet: before anything READS them
// (the one-sided kernels, the pilot, the exact kernel; the self-join form reads the index's own copy in position order).
int sg_csr_ensure_rows(sg_ctx *ctx, const sg_csr *m) {
    if (m && m->rows_of && m->rows_of->pending_src) return sg_collapse_materialize(ctx, m->rows_of);
    return SG_OK;
}

void sg_collapse_free(SgCollapse *c) {
    if (!c) return;
    sg_ctx *ctx = c->ctx;
    ctx->release(c->d_gid);
    ctx->release(c->d_group_ptr);
    ctx->release(c->d_members);
    ctx->release(c->d_rep_rows);
    ctx->release(c->d_rep_start);
    ctx->release(c->d_rep_len);
    sg_csr_free(c->unique);
    delete c;
}

static int collapse_groups(sg_ctx *ctx, const sg_csr *B, bool forced, bool by_table, bool defer_rows, SgCollapse **out);

// *out stays null when collapsing is off, not worth it (fewer than 3 % repeats) or not possible.
// left_side: the groups are those of a LEFT matrix of a one-sided product (sg_spgemm_topn): its own switch
// (SG_COLLAPSE_LEFT=0 off, =1 from two rows on) and a higher bar by default -- the grouping is paid by the multiply that
// asks for it, not by an index build that many multiplies share.
int sg_collapse_build(sg_ctx *ctx, const sg_csr *B, SgCollapse **out, bool left_side, bool defer_rows) {
    *out = nullptr;
    const char *sw = ctx->opt("SG_COLLAPSE");
    if ((sw && sw[0] == '0') || B->n_rows < 2 || B->nnz <= 0 || B->n_rows >= ((int64_t)1 << 31)) return SG_OK;
    bool forced = sw && sw[0] == '1';
    int64_t min_rows = 8192;
    if (left_side) {
        const char *ls = ctx->opt("SG_COLLAPSE_LEFT");
        if (ls && ls[0] == '0') return SG_OK;
        forced = ls && ls[0] == '1';
        min_rows = 65536;
    }
    if (!forced && B->n_rows < min_rows) return SG_OK;
    const bool want_table = !(ctx->opt("SG_GROUP_SORT") && ctx->opt("SG_GROUP_SORT")[0] == '1');   // (=1: the sort-based path)
    int st = collapse_groups(ctx, B, forced, want_table, defer_rows, out);
    if (st == SG_OK && *out == nullptr && want_table && ctx->group_table_overflow) {
        // a group of more than SG_GROUP_SORT_LDS members (a hub of identical names): the sort-based path lists any group
        ctx->group_table_overflow = false;
        st = collapse_groups(ctx, B, forced, false, defer_rows, out);
    }
    return st;
}

// the representatives' matrix of the groups `c` (n_u rows, nnz_u entries): arrays allocated, rows written unless deferred
static int collapse_unique_matrix(sg_ctx *ctx, const sg_csr *B, SgCollapse *c, int64_t nnz_u, int64_t *ptr /* given: filled already */,
                                  bool defer_rows) {
    const int64_t n_u = c->n_u;
    int32_t *idx = nullptr;
    void *val = nullptr;
    const size_t vs = B->dtype == SG_F64 ? 8 : 4;
    int st = SG_OK;
    if (!ptr) st = sg_alloc(ctx, (size_t)n_u + 2, &ptr);
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)nnz_u + 64, &idx);
    if (st == SG_OK) st = ctx->alloc(((size_t)nnz_u + 64) * vs, &val);
    sg_csr *m = st == SG_OK ? new (std::nothrow) sg_csr() : nullptr;
    if (st == SG_OK && !m) st = SG_ERR_OOM;
    if (st != SG_OK) {
        ctx->release(ptr);
        ctx->release(idx);
        ctx->release(val);
        return st;
    }
    m->ctx = ctx;
    m->n_rows = n_u;
    m->n_cols = B->n_cols;
    m->nnz = nnz_u;
    m->dtype = B->dtype;
    m->d_indptr = ptr;
    m->d_indices = idx;
    m->d_data = val;
    m->owned = true;
    m->props_state = B->props_state;          // a subset of B's rows: cosine-like if B is; the maxima are upper bounds
    m->props_max_norm2 = B->props_max_norm2;
    m->props_max_nnz = B->props_max_nnz;
    m->rows_of = c;
    c->unique = m;
    c->pending_src = B;       // (written by sg_collapse_materialize: now, or when somebody asks sg_csr_ensure_rows)
    if (!defer_rows) return sg_collapse_materialize(ctx, c);
    return SG_OK;
}

// Grouping through the hash table (see group_rows_kernel).  ONE host round trip: the number of groups, the entries of
// their representatives and what the member sort found arrive together, after everything has been queued on upper bounds
// (n rows for n_u groups); rounds 4-5 stopped twice.  *out stays null when grouping is not worth it, or when a group is too
// large for this path (ctx->group_table_overflow says so: the caller takes the sort-based one).
static int collapse_groups_table(sg_ctx *ctx, const sg_csr *B, bool forced, bool defer_rows, SgCollapse **out) {
    *out = nullptr;
    ctx->group_table_overflow = false;
    const int64_t n = B->n_rows;
    uint64_t table_size = 1024;
    while (table_size < 2 * (uint64_t)n) table_size <<= 1;
    unsigned long long *table = nullptr;
    uint32_t *slot_of_row = nullptr, *rep_of_row = nullptr, *rep_excl = nullptr, *size = nullptr, *cursor = nullptr, *queue = nullptr;
    uint32_t *totals = nullptr;   // [0] groups, [4 .. 36) the member sort's words, [64 .. 64 + 32 * 32) entries of the representatives (32 partial sums of 64 bits, 128 bytes apart)
    SgCollapse *c = new (std::nothrow) SgCollapse();
    if (!c) return SG_ERR_OOM;
    c->ctx = ctx;
    c->n_orig = n;
    int st = sg_alloc(ctx, (size_t)table_size, &table);
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)n + 1, &slot_of_row);
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)n + 1, &rep_of_row);
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)n + 1, &rep_excl);
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)n + 2, &size);
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)n + 2, &cursor);
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)n + 2, &queue);
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)(64 + 32 * 32), &totals);
    // (sized for n groups: the number is not known to the host while these are queued)
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)n + 1, &c->d_gid);
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)n + 2, &c->d_group_ptr);
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)n + 1, &c->d_members);
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)n + 1, &c->d_rep_rows);
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)n + 2, &c->d_rep_start);
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)n + 2, &c->d_rep_len);
    auto cleanup = [&]() {
        ctx->release(table);
        ctx->release(slot_of_row);
        ctx->release(rep_of_row);
        ctx->release(rep_excl);
        ctx->release(size);
        ctx->release(cursor);
        ctx->release(queue);
        ctx->release(totals);
    };
    const unsigned g1 = (unsigned)((n + 255) / 256), g16 = (unsigned)((((n + SG_GROUP_ROWS - 1) / SG_GROUP_ROWS) * 16 + 255) / 256);
    if (st == SG_OK)
        st = SG_ZERO4(ctx, table, sizeof(unsigned long long) * (size_t)table_size, size, sizeof(uint32_t) * (size_t)(n + 2), cursor,
                      sizeof(uint32_t) * (size_t)(n + 2), totals, (64 + 32 * 32) * sizeof(uint32_t));
    if (st == SG_OK) {
        if (B->dtype == SG_F64)
            hipLaunchKernelGGL(group_rows_kernel<double>, dim3(g16), dim3(256), 0, ctx->stream, B->d_indptr, B->d_indices,
                               (const double *)B->d_data, n, table, (uint32_t)(table_size - 1), slot_of_row);
        else
            hipLaunchKernelGGL(group_rows_kernel<float>, dim3(g16), dim3(256), 0, ctx->stream, B->d_indptr, B->d_indices,
                               (const float *)B->d_data, n, table, (uint32_t)(table_size - 1), slot_of_row);
        // representatives flagged and counted by the scan that needs the flags (totals[0] = number of groups)
        st = sg_scan_launch<uint32_t>(ctx, GroupRepLoad{table, slot_of_row, rep_of_row}, SgScanStoreArray<uint32_t>{rep_excl}, n, totals);
    }
    if (st == SG_OK) {
        hipLaunchKernelGGL(group_ids_kernel, dim3(g1), dim3(256), 0, ctx->stream, (const uint32_t *)rep_of_row, (const uint32_t *)rep_excl,
                           (const uint32_t *)nullptr, n, c->d_gid, size, c->d_rep_rows, B->d_indptr, c->d_rep_start, c->d_rep_len,
                           (unsigned long long *)(totals + 64));
        // (over n + 1 sizes, zeros behind the last group: group_ptr[n_u] = n comes out by itself)
        st = sg_exclusive_scan_u32(ctx, size, c->d_group_ptr, n + 1, nullptr);
    }
    if (st == SG_OK) {
        hipLaunchKernelGGL(group_scatter_kernel, dim3(g1), dim3(256), 0, ctx->stream, (const uint32_t *)c->d_gid,
                           (const uint32_t *)c->d_group_ptr, n, cursor, c->d_members);
        hipLaunchKernelGGL(group_sort_small_kernel, dim3(g1), dim3(256), 0, ctx->stream, (const uint32_t *)c->d_group_ptr,
                           (const uint32_t *)totals, c->d_members, totals + 4, queue);
        hipLaunchKernelGGL(group_sort_lds_kernel, dim3(512), dim3(256), 0, ctx->stream, (const uint32_t *)c->d_group_ptr, c->d_members,
                           (const uint32_t *)(totals + 4), (const uint32_t *)queue);
        if (hipGetLastError() != hipSuccess) st = SG_ERR_HIP;
    }
    uint32_t h[64 + 32 * 32];
    for (auto &w : h) w = 0;
    static_assert(sizeof(h) <= SG_H_FETCH_WORDS * sizeof(uint32_t), "the pinned read-back buffer holds the grouping's words");
    if (st == SG_OK) {   // (through pinned memory: see sg_ctx::h_fetch)
        if (hipMemcpyAsync(ctx->h_fetch, totals, sizeof(h), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess)
            st = SG_ERR_HIP;
        else memcpy(h, ctx->h_fetch, sizeof(h));
    }
    const uint32_t n_groups = h[0];
    int64_t nnz_u = 0;
    for (int q = 0; q < 32; ++q) nnz_u += (int64_t)(((uint64_t)h[64 + 32 * q + 1] << 32) | (uint64_t)h[64 + 32 * q]);
    const uint32_t *h_words = h + 4;      // [0] groups queued for the LDS sort, [1] the largest group, [2] very large groups, [3 ..] which
    if (st != SG_OK || n_groups == 0 || (!forced && (double)n_groups > 0.97 * (double)n) || (int64_t)n_groups == n) {
        cleanup();
        sg_collapse_free(c);
        return st;
    }
    if (h_words[2] > SG_GROUP_LARGE_MAX) {
        // dozens of very large groups: the sort-based path lists any number of them in one go -- the caller takes it
        ctx->group_table_overflow = true;
        cleanup();
        sg_collapse_free(c);
        return SG_OK;
    }
    c->n_u = n_groups;
    for (uint32_t q = 0; q < h_words[2] && st == SG_OK; ++q) {
        // a group too large for the workgroup sort: flag / prefix sum / scatter of its own (size and rep_excl have served)
        const uint32_t g = h_words[3 + q];
        hipLaunchKernelGGL(large_group_flag_kernel, dim3(g1), dim3(256), 0, ctx->stream, (const uint32_t *)c->d_gid, n, g, size);
        st = sg_exclusive_scan_u32(ctx, size, rep_excl, n, nullptr);
        if (st == SG_OK) {
            hipLaunchKernelGGL(large_group_fill_kernel, dim3(g1), dim3(256), 0, ctx->stream, (const uint32_t *)c->d_gid,
                               (const uint32_t *)rep_excl, n, g, (const uint32_t *)c->d_group_ptr, c->d_members);
            if (hipGetLastError() != hipSuccess) st = SG_ERR_HIP;
        }
    }
    cleanup();
    if (st == SG_OK) st = collapse_unique_matrix(ctx, B, c, nnz_u, nullptr, defer_rows);
    if (st != SG_OK) {
        sg_collapse_free(c);
        return st;
    }
    *out = c;
    return SG_OK;
}

// The sort-based way to the groups (lists with dozens of very large groups; SG_GROUP_SORT=1); *out stays null when grouping
// is not worth it.
static int collapse_groups(sg_ctx *ctx, const sg_csr *B, bool forced, bool by_table, bool defer_rows, SgCollapse **out) {
    if (by_table) return collapse_groups_table(ctx, B, forced, defer_rows, out);
    *out = nullptr;
    ctx->group_table_overflow = false;
    const int64_t n = B->n_rows;
    uint64_t *hash = nullptr, *hash_sorted = nullptr;
    uint32_t *row_id = nullptr, *row_sorted = nullptr, *head = nullptr, *run_excl = nullptr, *head_pos = nullptr;
    uint32_t *rep_of_row = nullptr, *rank_of_row = nullptr, *is_rep = nullptr, *rep_excl = nullptr, *size = nullptr;
    uint32_t *totals = nullptr;
    SgCollapse *c = nullptr;
    int st = sg_alloc(ctx, (size_t)n + 1, &hash);
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)40, &totals);
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)n + 1, &hash_sorted);
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)n + 1, &row_id);
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)n + 1, &row_sorted);
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)n + 1, &head);
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)n + 1, &run_excl);
    auto cleanup = [&]() {
        ctx->release(hash);
        ctx->release(hash_sorted);
        ctx->release(row_id);
        ctx->release(row_sorted);
        ctx->release(head);
        ctx->release(run_excl);
        ctx->release(head_pos);
        ctx->release(rep_of_row);
        ctx->release(rank_of_row);
        ctx->release(is_rep);
        ctx->release(rep_excl);
        ctx->release(size);
        ctx->release(totals);
    };
    const unsigned g1 = (unsigned)((n + 255) / 256), g16 = (unsigned)((n * 16 + 255) / 256);
    if (st == SG_OK) {
        if (B->dtype == SG_F64)
            hipLaunchKernelGGL(row_hash_kernel<double>, dim3(g16), dim3(256), 0, ctx->stream, B->d_indptr, B->d_indices,
                               (const double *)B->d_data, n, hash, row_id);
        else
            hipLaunchKernelGGL(row_hash_kernel<float>, dim3(g16), dim3(256), 0, ctx->stream, B->d_indptr, B->d_indices,
                               (const float *)B->d_data, n, hash, row_id);
        st = sg_sort_pairs_u64_u32(ctx, hash, row_id, n, hash_sorted, row_sorted);
    }
    if (st == SG_OK) {
        if (B->dtype == SG_F64)
            hipLaunchKernelGGL(group_heads_kernel<double>, dim3(g1), dim3(256), 0, ctx->stream, B->d_indptr, B->d_indices,
                               (const double *)B->d_data, n, (const uint64_t *)hash_sorted, (const uint32_t *)row_sorted, head);
        else
            hipLaunchKernelGGL(group_heads_kernel<float>, dim3(g1), dim3(256), 0, ctx->stream, B->d_indptr, B->d_indices,
                               (const float *)B->d_data, n, (const uint64_t *)hash_sorted, (const uint32_t *)row_sorted, head);
        st = sg_exclusive_scan_u32(ctx, head, run_excl, n, totals);   // totals[0] = number of groups
    }
    uint32_t n_groups = 0;
    if (st == SG_OK) {
        if (hipMemcpyAsync(&n_groups, totals, 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess)
            st = SG_ERR_HIP;
    }
    if (st != SG_OK || n_groups == 0 || (!forced && (double)n_groups > 0.97 * (double)n) || (int64_t)n_groups == n) {
        cleanup();
        return st;
    }
    const int64_t n_u = n_groups;
    c = new (std::nothrow) SgCollapse();
    if (!c) {
        cleanup();
        return SG_ERR_OOM;
    }
    c->ctx = ctx;
    c->n_orig = n;
    c->n_u = n_u;
    st = sg_alloc(ctx, (size_t)n_u + 1, &size);
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)n + 1, &c->d_gid);
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)n_u + 2, &c->d_group_ptr);
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)n + 1, &c->d_members);
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)n_u + 1, &c->d_rep_rows);
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)n_u + 1, &head_pos);
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)n + 1, &rep_of_row);
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)n + 1, &rank_of_row);
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)n + 1, &is_rep);
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)n + 1, &rep_excl);
    if (st == SG_OK) st = SG_ZERO2(ctx, is_rep, sizeof(uint32_t) * (size_t)(n + 1), size, sizeof(uint32_t) * (size_t)(n_u + 1));
    if (st == SG_OK) {
        hipLaunchKernelGGL(head_pos_kernel, dim3(g1), dim3(256), 0, ctx->stream, (const uint32_t *)head, (const uint32_t *)run_excl, n,
                           head_pos);
        hipLaunchKernelGGL(group_members_kernel, dim3(g1), dim3(256), 0, ctx->stream, (const uint32_t *)head,
                           (const uint32_t *)run_excl, (const uint32_t *)head_pos, (const uint32_t *)row_sorted, n, rep_of_row,
                           rank_of_row, is_rep);
        st = sg_exclusive_scan_u32(ctx, is_rep, rep_excl, n, nullptr);
    }
    if (st == SG_OK) {
        hipLaunchKernelGGL(group_ids_kernel, dim3(g1), dim3(256), 0, ctx->stream, (const uint32_t *)rep_of_row,
                           (const uint32_t *)rep_excl, (const uint32_t *)is_rep, n, c->d_gid, size, c->d_rep_rows,
                           (const int64_t *)nullptr, (int64_t *)nullptr, (int32_t *)nullptr, (unsigned long long *)nullptr);
        st = sg_exclusive_scan_u32(ctx, size, c->d_group_ptr, n_u, c->d_group_ptr + n_u);
    }
    if (st == SG_OK) {
        hipLaunchKernelGGL(group_fill_kernel, dim3(g1), dim3(256), 0, ctx->stream, (const uint32_t *)c->d_gid,
                           (const uint32_t *)rank_of_row, (const uint32_t *)c->d_group_ptr, n, c->d_members);
        if (hipGetLastError() != hipSuccess) st = SG_ERR_HIP;
    }
    // the matrix of the representatives: its row pointers here (one more round trip for the number of entries)
    int32_t *len = nullptr;
    int64_t *ptr = nullptr;
    int64_t nnz_u = 0;
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)n_u + 1, &len);
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)n_u + 2, &ptr);
    if (st == SG_OK) {
        hipLaunchKernelGGL(unique_len_kernel, dim3((unsigned)((n_u + 255) / 256)), dim3(256), 0, ctx->stream, B->d_indptr,
                           (const uint32_t *)c->d_rep_rows, n_u, len);
        st = sg_exclusive_scan_i32_to_i64(ctx, len, ptr, n_u);
    }
    if (st == SG_OK && (hipMemcpyAsync(&nnz_u, ptr + n_u, 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                        hipStreamSynchronize(ctx->stream) != hipSuccess))
        st = SG_ERR_HIP;
    ctx->release(len);
    cleanup();
    if (st == SG_OK) {
        st = collapse_unique_matrix(ctx, B, c, nnz_u, ptr, defer_rows);
        ptr = nullptr;    // (the matrix's, or released by collapse_unique_matrix)
    }
    ctx->release(ptr);
    if (st != SG_OK) {
        sg_collapse_free(c);
        return st;
    }
    *out = c;
    return SG_OK;
}

// ------------------------------------------------------------------------------------------------ expansion
// Thread per output row: the row's groups written out member by member; rows in which a group of several members ties
// with another group are queued for the wave-per-row kernel.
template <typename T>
__global__ void __launch_bounds__(256) expand_simple_kernel(const int32_t *__restrict__ u_cols, const T *__restrict__ u_vals,
                                                            const int32_t *__restrict__ u_cnt, int32_t u_stride,
                                                            const uint32_t *__restrict__ gid /* null: output row r = row r of u */,
                                                            const int32_t *__restrict__ row_list /* null: output row r is row r */,
                                                            const uint32_t *__restrict__ group_ptr, const uint32_t *__restrict__ members,
                                                            int64_t n_out, int32_t stride, int32_t *__restrict__ cols,
                                                            T *__restrict__ vals, int32_t *__restrict__ cnt,
                                                            uint32_t *__restrict__ slow_count, uint32_t *__restrict__ slow_rows) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_out) return;
    const int64_t row = row_list ? (int64_t)row_list[r] : r;
    const int64_t ur = gid ? (int64_t)gid[row] : row;
    const int32_t m = u_cnt[ur];
    const int32_t *uc = u_cols + ur * u_stride;
    const T *uv = u_vals + ur * u_stride;
    // A group of several members whose score no other group of the row shares is written member by member (members
    // ascend); groups of one member that share a score are in column order already (groups are numbered by ascending
    // lowest member, and the row over groups is sorted by score, then group).  Only a group of several members that
    // TIES with another group needs the merge by column: the wave-per-row kernel rewrites such rows from scratch.
    int32_t out = 0;
    T prev = (T)0;
    for (int32_t e = 0; e < m && out < stride; ++e) {
        const uint32_t lo = group_ptr[uc[e]], hi = group_ptr[uc[e] + 1];
        const T v = uv[e];
        if (hi - lo != 1u && ((e > 0 && prev == v) || (e + 1 < m && uv[e + 1] == v))) {
            slow_rows[atomicAdd(slow_count, 1u)] = (uint32_t)r;
            return;
        }
        for (uint32_t p = lo; p < hi && out < stride; ++p, ++out) {
            cols[r * stride + out] = (int32_t)members[p];
            vals[r * stride + out] = v;
        }
        prev = v;
    }
    cnt[r] = out;
}

// Wave per queued row: the groups of one score are merged by column (every lane holds one group's next member, the
// smallest of all lanes is written, that lane advances), score after score, until top_n columns are out.
template <typename T>
__global__ void __launch_bounds__(64) expand_merge_kernel(const int32_t *__restrict__ u_cols, const T *__restrict__ u_vals,
                                                          const int32_t *__restrict__ u_cnt, int32_t u_stride,
                                                          const uint32_t *__restrict__ gid, const int32_t *__restrict__ row_list,
                                                          const uint32_t *__restrict__ group_ptr,
                                                          const uint32_t *__restrict__ members, int32_t stride,
                                                          int32_t *__restrict__ cols, T *__restrict__ vals, int32_t *__restrict__ cnt,
                                                          const uint32_t *__restrict__ slow_count, const uint32_t *__restrict__ slow_rows) {
    const int lane = threadIdx.x;
    const uint32_t n_slow = *slow_count;
    for (uint32_t q = blockIdx.x; q < n_slow; q += gridDim.x) {
        const int64_t r = slow_rows[q];
        const int64_t row = row_list ? (int64_t)row_list[r] : r;
        const int64_t ur = gid ? (int64_t)gid[row] : row;
        const int32_t m = u_cnt[ur];
        const int32_t *uc = u_cols + ur * u_stride;
        const T *uv = u_vals + ur * u_stride;
        int32_t out = 0;
        int32_t e0 = 0;
        while (e0 < m && out < stride) {
            const T score = uv[e0];
            int32_t e1 = e0 + 1;
            while (e1 < m && uv[e1] == score) ++e1;      // groups [e0, e1) share the score
            // lanes take the groups of the run 64 at a time; a run longer than 64 groups keeps, per lane, the group whose
            // next member is smallest among the lane's groups -- re-evaluated after every output (rare: equal scores)
            const int32_t run = e1 - e0;
            if (run <= 64) {
                uint32_t cur = 0, end = 0;
                if (lane < run) {
                    cur = group_ptr[uc[e0 + lane]];
                    end = group_ptr[uc[e0 + lane] + 1];
                }
                uint32_t nxt = (lane < run && cur < end) ? members[cur] : 0xFFFFFFFFu;
                while (out < stride) {
                    uint32_t best = nxt;
#pragma unroll
                    for (int d = 32; d > 0; d >>= 1) best = min(best, (uint32_t)__shfl_xor((int)best, d, 64));
                    if (best == 0xFFFFFFFFu) break;
                    if (lane == 0) {
                        cols[r * stride + out] = (int32_t)best;
                        vals[r * stride + out] = score;
                    }
                    ++out;
                    if (nxt == best) {   // (members are distinct rows: exactly one lane)
                        ++cur;
                        nxt = cur < end ? members[cur] : 0xFFFFFFFFu;
                    }
                }
            } else {
                // every output: each lane scans its share of the run's groups for the smallest member not yet written
                // (members ascend within a group: binary position = members written so far from it is not kept, so the
                // bound is the last column written)
                uint32_t last = 0;
                bool any_written = false;
                while (out < stride) {
                    uint32_t best = 0xFFFFFFFFu;
                    for (int32_t g = lane; g < run; g += 64) {
                        uint32_t lo = group_ptr[uc[e0 + g]], hi = group_ptr[uc[e0 + g] + 1];
                        // first member > last (or the first member at all)
                        while (lo < hi) {
                            const uint32_t mid = (lo + hi) >> 1;
                            if (any_written && members[mid] <= last) lo = mid + 1;
                            else hi = mid;
                        }
                        if (lo < group_ptr[uc[e0 + g] + 1]) best = min(best, members[lo]);
                    }
#pragma unroll
                    for (int d = 32; d > 0; d >>= 1) best = min(best, (uint32_t)__shfl_xor((int)best, d, 64));
                    if (best == 0xFFFFFFFFu) break;
                    if (lane == 0) {
                        cols[r * stride + out] = (int32_t)best;
                        vals[r * stride + out] = score;
                    }
                    ++out;
                    last = best;
                    any_written = true;
                }
            }
            e0 = e1;
        }
        if (lane == 0) cnt[r] = out;
    }
}

// ru: result over groups (rows: groups if `rows_are_groups`, else the caller's left rows).  out: allocated by the caller
// (n_out rows, stride), filled here.
int sg_collapse_expand(sg_ctx *ctx, const SgCollapse *c, const sg_topn *ru, bool rows_are_groups, sg_topn *out,
                       const int32_t *row_list) {
    const int64_t n_out = out->n_rows;
    if (n_out <= 0) return SG_OK;
    uint32_t *slow = nullptr;
    SG_TRY(sg_alloc(ctx, (size_t)n_out + 4, &slow));
    // (the output's counts and the slow-row queue's head in one launch; the callers do not clear the counts themselves)
    int st = SG_ZERO2(ctx, out->d_counts, sizeof(int32_t) * (size_t)(n_out + 1), slow, 16);
    if (st == SG_OK) {
        // (tried in round 6: sixteen lanes per output row, contiguous reads and writes -- 0.82 instead of 0.64 ms at 5 M: a
        //  chain of four dependent loads per row with one row per sixteen lanes in flight; a thread per row keeps ten going)
        const unsigned g1 = (unsigned)((n_out + 255) / 256);
        const uint32_t *gid = rows_are_groups ? c->d_gid : nullptr;
        if (out->dtype == SG_F64) {
            hipLaunchKernelGGL(expand_simple_kernel<double>, dim3(g1), dim3(256), 0, ctx->stream, (const int32_t *)ru->d_cols,
                               (const double *)ru->d_vals, (const int32_t *)ru->d_counts, ru->stride, gid, row_list,
                               (const uint32_t *)c->d_group_ptr, (const uint32_t *)c->d_members, n_out, out->stride, out->d_cols,
                               (double *)out->d_vals, out->d_counts, slow, slow + 4);
            hipLaunchKernelGGL(expand_merge_kernel<double>, dim3(2048), dim3(64), 0, ctx->stream, (const int32_t *)ru->d_cols,
                               (const double *)ru->d_vals, (const int32_t *)ru->d_counts, ru->stride, gid, row_list,
                               (const uint32_t *)c->d_group_ptr, (const uint32_t *)c->d_members, out->stride, out->d_cols,
                               (double *)out->d_vals, out->d_counts, (const uint32_t *)slow, (const uint32_t *)(slow + 4));
        } else {
            hipLaunchKernelGGL(expand_simple_kernel<float>, dim3(g1), dim3(256), 0, ctx->stream, (const int32_t *)ru->d_cols,
                               (const float *)ru->d_vals, (const int32_t *)ru->d_counts, ru->stride, gid, row_list,
                               (const uint32_t *)c->d_group_ptr, (const uint32_t *)c->d_members, n_out, out->stride, out->d_cols,
                               (float *)out->d_vals, out->d_counts, slow, slow + 4);
            hipLaunchKernelGGL(expand_merge_kernel<float>, dim3(2048), dim3(64), 0, ctx->stream, (const int32_t *)ru->d_cols,
                               (const float *)ru->d_vals, (const int32_t *)ru->d_counts, ru->stride, gid, row_list,
                               (const uint32_t *)c->d_group_ptr, (const uint32_t *)c->d_members, out->stride, out->d_cols,
                               (float *)out->d_vals, out->d_counts, (const uint32_t *)slow, (const uint32_t *)(slow + 4));
        }
        if (hipGetLastError() != hipSuccess) st = SG_ERR_HIP;
    }
    ctx->release(slow);
    return st;
}
