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

template <typename T>
__global__ void __launch_bounds__(256) row_hash_kernel(const int64_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                                                       const T *__restrict__ data, int64_t n_rows, uint64_t *__restrict__ hash,
                                                       uint32_t *__restrict__ row_id /* null: not wanted */,
                                                       uint32_t *__restrict__ table /* null, or the slots to mark empty */,
                                                       uint64_t table_size) {
    const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const int sub = threadIdx.x & 15;
    if (table)   // (the table of the grouping below: cleared here instead of by a launch of its own)
        for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < table_size; i += (uint64_t)gridDim.x * blockDim.x)
            table[i] = 0xFFFFFFFFu;
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
#define SG_GROUP_EMPTY 0xFFFFFFFFu
#define SG_GROUP_SORT_LDS 8192
#define SG_GROUP_LARGE_MAX 28u      // very large groups listed per call (more: the sort-based path)
__global__ void __launch_bounds__(256) group_insert_kernel(const uint64_t *__restrict__ hash, int64_t n_rows, uint32_t *table,
                                                           uint32_t mask, uint32_t *__restrict__ slot_of_row) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    const uint64_t h = hash[r];
    uint32_t s = (uint32_t)(h ^ (h >> 29)) & mask;
    for (;;) {
        uint32_t cur = __hip_atomic_load(&table[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == SG_GROUP_EMPTY) {
            cur = atomicCAS(&table[s], SG_GROUP_EMPTY, (uint32_t)r);
            if (cur == SG_GROUP_EMPTY) break;                 // the slot is this hash's now
        }
        if (hash[cur] == h) {                                 // (whoever holds the slot has the slot's hash)
            // (a hub of 66 000 identical names is 66 000 rows at ONE slot: only a row below what the slot shows sends an
            //  atomic -- the value only ever falls, so a stale reading only costs an atomic that changes nothing)
            if ((uint32_t)r < cur) atomicMin(&table[s], (uint32_t)r);
            break;
        }
        s = (s + 1u) & mask;
    }
    slot_of_row[r] = s;
}

// rep_of_row[r] = the lowest row with r's content (r itself when it is that row, or when its hash's slot names a row of
// other content); is_rep[r] = 1 for representatives.  Sixteen lanes per row.
template <typename T>
__global__ void __launch_bounds__(256) group_verify_kernel(const int64_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                                                           const T *__restrict__ data, int64_t n_rows,
                                                           const uint32_t *__restrict__ table, const uint32_t *__restrict__ slot_of_row,
                                                           uint32_t *__restrict__ rep_of_row, uint32_t *__restrict__ is_rep) {
    const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const int sub = threadIdx.x & 15;
    if (r >= n_rows) return;
    const int64_t rep = table[slot_of_row[r]];
    bool same = true;
    if (rep != r) {
        const int64_t la = indptr[r], lb = indptr[rep];
        const int64_t n = indptr[r + 1] - la;
        same = n == indptr[rep + 1] - lb;
        if (same)
            for (int64_t e = sub; e < n; e += 16) same = same && indices[la + e] == indices[lb + e] && data[la + e] == data[lb + e];
#pragma unroll
        for (int d = 8; d > 0; d >>= 1) same = same && (__shfl_xor((int)same, d, 64) != 0);
    }
    if (sub == 0) {
        const bool own = rep == r || !same;
        rep_of_row[r] = own ? (uint32_t)r : (uint32_t)rep;
        is_rep[r] = own ? 1u : 0u;
    }
}

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
__global__ void __launch_bounds__(256) group_sort_small_kernel(const uint32_t *__restrict__ group_ptr, int64_t n_u, uint32_t *members,
                                                               uint32_t *words, uint32_t *__restrict__ queue) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_u) return;
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

// gid[row] = number of representatives below the row's representative; sizes counted; rep_rows[gid] = representative
__global__ void __launch_bounds__(256) group_ids_kernel(const uint32_t *__restrict__ rep_of_row, const uint32_t *__restrict__ rep_excl,
                                                        const uint32_t *__restrict__ is_rep, int64_t n_rows,
                                                        uint32_t *__restrict__ gid, uint32_t *__restrict__ size,
                                                        uint32_t *__restrict__ rep_rows) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = r < n_rows;
    uint32_t g = 0xFFFFFFFFu;
    bool rep = false;
    if (valid) {
        g = rep_excl[rep_of_row[r]];
        gid[r] = g;
        rep = is_rep[r] != 0u;
        if (rep) {
            rep_rows[g] = (uint32_t)r;
            atomicAdd(&size[g], 1u);          // (one representative per group: an address of its own)
        }
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

void sg_collapse_free(SgCollapse *c) {
    if (!c) return;
    sg_ctx *ctx = c->ctx;
    ctx->release(c->d_gid);
    ctx->release(c->d_group_ptr);
    ctx->release(c->d_members);
    ctx->release(c->d_rep_rows);
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

// One of the two ways to the groups; *out stays null when grouping is not worth it (or, table path, when a group is too
// large for it: ctx->group_table_overflow says so).
static int collapse_groups(sg_ctx *ctx, const sg_csr *B, bool forced, bool by_table, bool defer_rows, SgCollapse **out) {
    *out = nullptr;
    ctx->group_table_overflow = false;
    const int64_t n = B->n_rows;
    uint64_t *hash = nullptr, *hash_sorted = nullptr;
    uint32_t *row_id = nullptr, *row_sorted = nullptr, *head = nullptr, *run_excl = nullptr, *head_pos = nullptr;
    uint32_t *rep_of_row = nullptr, *rank_of_row = nullptr, *is_rep = nullptr, *rep_excl = nullptr, *size = nullptr;
    uint32_t *totals = nullptr, *table = nullptr, *slot_of_row = nullptr, *cursor = nullptr, *queue = nullptr;
    SgCollapse *c = nullptr;
    uint64_t table_size = 0;
    int st = sg_alloc(ctx, (size_t)n + 1, &hash);
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)40, &totals);
    if (by_table) {
        table_size = 1024;
        while (table_size < 2 * (uint64_t)n) table_size <<= 1;
        if (st == SG_OK) st = sg_alloc(ctx, (size_t)table_size, &table);
        if (st == SG_OK) st = sg_alloc(ctx, (size_t)n + 1, &slot_of_row);
        if (st == SG_OK) st = sg_alloc(ctx, (size_t)n + 1, &rep_of_row);
        if (st == SG_OK) st = sg_alloc(ctx, (size_t)n + 1, &is_rep);
        if (st == SG_OK) st = sg_alloc(ctx, (size_t)n + 1, &rep_excl);
    } else {
        if (st == SG_OK) st = sg_alloc(ctx, (size_t)n + 1, &hash_sorted);
        if (st == SG_OK) st = sg_alloc(ctx, (size_t)n + 1, &row_id);
        if (st == SG_OK) st = sg_alloc(ctx, (size_t)n + 1, &row_sorted);
        if (st == SG_OK) st = sg_alloc(ctx, (size_t)n + 1, &head);
        if (st == SG_OK) st = sg_alloc(ctx, (size_t)n + 1, &run_excl);
    }
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
        ctx->release(table);
        ctx->release(slot_of_row);
        ctx->release(cursor);
        ctx->release(queue);
    };
    const unsigned g1 = (unsigned)((n + 255) / 256), g16 = (unsigned)((n * 16 + 255) / 256);
    if (st == SG_OK) {
        if (B->dtype == SG_F64)
            hipLaunchKernelGGL(row_hash_kernel<double>, dim3(g16), dim3(256), 0, ctx->stream, B->d_indptr, B->d_indices,
                               (const double *)B->d_data, n, hash, row_id, table, table_size);
        else
            hipLaunchKernelGGL(row_hash_kernel<float>, dim3(g16), dim3(256), 0, ctx->stream, B->d_indptr, B->d_indices,
                               (const float *)B->d_data, n, hash, row_id, table, table_size);
    }
    if (st == SG_OK && by_table) {
        hipLaunchKernelGGL(group_insert_kernel, dim3(g1), dim3(256), 0, ctx->stream, (const uint64_t *)hash, n, table,
                           (uint32_t)(table_size - 1), slot_of_row);
        if (B->dtype == SG_F64)
            hipLaunchKernelGGL(group_verify_kernel<double>, dim3(g16), dim3(256), 0, ctx->stream, B->d_indptr, B->d_indices,
                               (const double *)B->d_data, n, (const uint32_t *)table, (const uint32_t *)slot_of_row, rep_of_row, is_rep);
        else
            hipLaunchKernelGGL(group_verify_kernel<float>, dim3(g16), dim3(256), 0, ctx->stream, B->d_indptr, B->d_indices,
                               (const float *)B->d_data, n, (const uint32_t *)table, (const uint32_t *)slot_of_row, rep_of_row, is_rep);
        st = sg_exclusive_scan_u32(ctx, is_rep, rep_excl, n, totals);   // totals[0] = number of groups
    } else if (st == SG_OK) {
        st = sg_sort_pairs_u64_u32(ctx, hash, row_id, n, hash_sorted, row_sorted);
        if (st == SG_OK) {
            if (B->dtype == SG_F64)
                hipLaunchKernelGGL(group_heads_kernel<double>, dim3(g1), dim3(256), 0, ctx->stream, B->d_indptr, B->d_indices,
                                   (const double *)B->d_data, n, (const uint64_t *)hash_sorted, (const uint32_t *)row_sorted, head);
            else
                hipLaunchKernelGGL(group_heads_kernel<float>, dim3(g1), dim3(256), 0, ctx->stream, B->d_indptr, B->d_indices,
                                   (const float *)B->d_data, n, (const uint64_t *)hash_sorted, (const uint32_t *)row_sorted, head);
            st = sg_exclusive_scan_u32(ctx, head, run_excl, n, totals);   // totals[0] = number of groups
        }
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
    if (by_table) {
        if (st == SG_OK) st = sg_alloc(ctx, (size_t)n_u + 1, &cursor);
        if (st == SG_OK) st = sg_alloc(ctx, (size_t)n_u + 1, &queue);
        if (st == SG_OK)
            st = SG_ZERO3(ctx, size, sizeof(uint32_t) * (size_t)(n_u + 1), cursor, sizeof(uint32_t) * (size_t)(n_u + 1), totals + 4,
                          32 * sizeof(uint32_t));
    } else {
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
    }
    if (st == SG_OK) {
        hipLaunchKernelGGL(group_ids_kernel, dim3(g1), dim3(256), 0, ctx->stream, (const uint32_t *)rep_of_row,
                           (const uint32_t *)rep_excl, (const uint32_t *)is_rep, n, c->d_gid, size, c->d_rep_rows);
        st = sg_exclusive_scan_u32(ctx, size, c->d_group_ptr, n_u, c->d_group_ptr + n_u);
    }
    if (st == SG_OK && by_table) {
        const unsigned gu1 = (unsigned)((n_u + 255) / 256);
        hipLaunchKernelGGL(group_scatter_kernel, dim3(g1), dim3(256), 0, ctx->stream, (const uint32_t *)c->d_gid,
                           (const uint32_t *)c->d_group_ptr, n, cursor, c->d_members);
        hipLaunchKernelGGL(group_sort_small_kernel, dim3(gu1), dim3(256), 0, ctx->stream, (const uint32_t *)c->d_group_ptr, n_u,
                           c->d_members, totals + 4, queue);
        hipLaunchKernelGGL(group_sort_lds_kernel, dim3(512), dim3(256), 0, ctx->stream, (const uint32_t *)c->d_group_ptr, c->d_members,
                           (const uint32_t *)(totals + 4), (const uint32_t *)queue);
        if (hipGetLastError() != hipSuccess) st = SG_ERR_HIP;
    } else if (st == SG_OK) {
        hipLaunchKernelGGL(group_fill_kernel, dim3(g1), dim3(256), 0, ctx->stream, (const uint32_t *)c->d_gid,
                           (const uint32_t *)rank_of_row, (const uint32_t *)c->d_group_ptr, n, c->d_members);
        if (hipGetLastError() != hipSuccess) st = SG_ERR_HIP;
    }
    // the matrix of the representatives
    int32_t *len = nullptr;
    int64_t *ptr = nullptr;
    int32_t *idx = nullptr;
    void *val = nullptr;
    const size_t vs = B->dtype == SG_F64 ? 8 : 4;
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)n_u + 1, &len);
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)n_u + 2, &ptr);
    if (st == SG_OK) st = sg_alloc(ctx, (size_t)B->nnz + 64, &idx);
    if (st == SG_OK) st = ctx->alloc(((size_t)B->nnz + 64) * vs, &val);
    int64_t nnz_u = 0;
    if (st == SG_OK) {
        hipLaunchKernelGGL(unique_len_kernel, dim3((unsigned)((n_u + 255) / 256)), dim3(256), 0, ctx->stream, B->d_indptr,
                           (const uint32_t *)c->d_rep_rows, n_u, len);
        st = sg_exclusive_scan_i32_to_i64(ctx, len, ptr, n_u);
    }
    if (st == SG_OK) {
        const unsigned gu = (unsigned)((n_u * 16 + 255) / 256);
        if (defer_rows)
            c->pending_src = B;       // (written by the index build, or by sg_collapse_materialize)
        else if (B->dtype == SG_F64)
            hipLaunchKernelGGL(unique_rows_kernel<double>, dim3(gu), dim3(256), 0, ctx->stream, B->d_indptr, B->d_indices,
                               (const double *)B->d_data, (const uint32_t *)c->d_rep_rows, n_u, (const int64_t *)ptr, idx,
                               (double *)val);
        else
            hipLaunchKernelGGL(unique_rows_kernel<float>, dim3(gu), dim3(256), 0, ctx->stream, B->d_indptr, B->d_indices,
                               (const float *)B->d_data, (const uint32_t *)c->d_rep_rows, n_u, (const int64_t *)ptr, idx, (float *)val);
        uint32_t h_words[32];      // table path: [0] groups queued for the LDS sort, [1] the largest group, [2] very large groups, [3 ..] which
        for (auto &w : h_words) w = 0;
        if (hipMemcpyAsync(&nnz_u, ptr + n_u, 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            (by_table && hipMemcpyAsync(h_words, totals + 4, sizeof(h_words), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) ||
            hipStreamSynchronize(ctx->stream) != hipSuccess)
            st = SG_ERR_HIP;
        if (st == SG_OK && by_table && h_words[2] > SG_GROUP_LARGE_MAX) {
            // dozens of very large groups: the sort-based path lists any number of them in one go -- the caller takes it
            ctx->group_table_overflow = true;
            ctx->release(len);
            ctx->release(ptr);
            ctx->release(idx);
            ctx->release(val);
            cleanup();
            sg_collapse_free(c);
            return SG_OK;
        }
        for (uint32_t q = 0; q < h_words[2] && st == SG_OK && by_table; ++q) {
            // (is_rep / rep_excl have served: flag and positions of the group's rows)
            const uint32_t g = h_words[3 + q];
            hipLaunchKernelGGL(large_group_flag_kernel, dim3(g1), dim3(256), 0, ctx->stream, (const uint32_t *)c->d_gid, n, g, is_rep);
            st = sg_exclusive_scan_u32(ctx, is_rep, rep_excl, n, nullptr);
            if (st == SG_OK) {
                hipLaunchKernelGGL(large_group_fill_kernel, dim3(g1), dim3(256), 0, ctx->stream, (const uint32_t *)c->d_gid,
                                   (const uint32_t *)rep_excl, n, g, (const uint32_t *)c->d_group_ptr, c->d_members);
                if (hipGetLastError() != hipSuccess) st = SG_ERR_HIP;
            }
        }
    }
    ctx->release(len);
    cleanup();
    sg_csr *m = st == SG_OK ? new (std::nothrow) sg_csr() : nullptr;
    if (st == SG_OK && !m) st = SG_ERR_OOM;
    if (st != SG_OK) {
        ctx->release(ptr);
        ctx->release(idx);
        ctx->release(val);
        sg_collapse_free(c);
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
    c->unique = m;
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
