"""numpy / oracle double of ``string_grouper_amd.distributed.HipOps`` for the world_size-2 gloo tests on CPU: same
methods, CPU torch tensors, the oracle's arithmetic.  TEST INFRASTRUCTURE ONLY.  The code under test is the
orchestration in string_grouper_amd/distributed.py (sharding, the df all-reduce, the ragged all-gathers, the
concatenation), which is the same code the GPU path runs."""
import numpy as np
import scipy.sparse as sp
import torch

from oracle import oracle as O
from oracle import port as P

KEY_BITS = 7
NGRAM = 3
KEY_SPACE = 1 << (KEY_BITS * NGRAM)


def _keys_of(strings):
    """Per string the packed keys of its 3-grams (the device's coding: 7 bits per character, big-endian)."""
    out = []
    for s in strings:
        grams = O.ngrams(s, NGRAM)
        out.append(np.array([(ord(g[0]) << 14) | (ord(g[1]) << 7) | ord(g[2]) for g in grams], dtype=np.int64))
    return out


class _State:
    def __init__(self, dtype):
        self.dtype = dtype
        self.df = torch.zeros(KEY_SPACE, dtype=torch.int32)
        self.keys_of_set = {}
        self.key_to_col = None
        self.idf = None


class NumpyOps:
    device = torch.device("cpu")

    def __init__(self, dtype=np.float32):
        self.dtype = dtype

    def n_strings(self, strings):
        return len(strings)

    def fit_begin(self, local_sets):
        st = _State(self.dtype)
        df = np.zeros(KEY_SPACE, np.int32)
        for s in local_sets:
            ks = _keys_of(s)
            st.keys_of_set[id(s)] = ks
            for k in ks:
                df[np.unique(k)] += 1
        st.df = torch.from_numpy(df)
        return st

    def df_tensor(self, st):
        return st.df

    def df_shareable(self, st):
        return True

    # test hook: ranks for which fit_info reports a table that cannot be added to the others' (a block coded over the
    # alphabet of its own strings)
    not_shareable_on = ()

    def fit_info(self, st):
        import torch.distributed as dist
        rank = dist.get_rank() if dist.is_initialized() else 0
        return rank not in self.not_shareable_on, KEY_SPACE

    def fit_end(self, st, n_docs_total):
        df = st.df.numpy()
        present = np.flatnonzero(df > 0)
        st.key_to_col = np.full(KEY_SPACE, -1, np.int64)
        st.key_to_col[present] = np.arange(len(present))
        st.idf = O.idf_vector(df[present], n_docs_total, self.dtype)

    def transform(self, st, strings):
        ks = st.keys_of_set.get(id(strings)) or _keys_of(strings)
        indptr, indices, data = [0], [], []
        for k in ks:
            cols = st.key_to_col[k]
            cols = cols[cols >= 0]
            u, c = np.unique(cols, return_counts=True)
            indices.extend(u.tolist())
            data.extend(c.tolist())
            indptr.append(len(indices))
        X = sp.csr_matrix((np.asarray(data, dtype=self.dtype), np.asarray(indices, np.int32), np.asarray(indptr, np.int32)),
                          shape=(len(ks), len(st.idf)))
        return O.tfidf_weight_normalize(X, st.idf)

    def csr_tensors(self, m):
        return (torch.from_numpy(m.indptr.astype(np.int64)), torch.from_numpy(m.indices.astype(np.int32)),
                torch.from_numpy(np.ascontiguousarray(m.data)))

    def csr_shape(self, m):
        return m.shape

    def csr_from_tensors(self, indptr, indices, data, shape):
        return sp.csr_matrix((data.numpy(), indices.numpy(), indptr.numpy()), shape=shape)

    # The library builds its index over a fixed permutation of the rows (sg_postings.hip): position p holds row
    # orig_of[p]; the ranges of the self-join form are then ranges of POSITIONS.  ``permuted = True`` restates that
    # contract here (same formula), so that the driver's handling of it runs on two real ranks.
    permuted = False

    @staticmethod
    def permutation(n):
        import math
        mult = int(0.6180339887498949 * n) | 1
        while math.gcd(mult, n) != 1:
            mult += 2
        pos_of = (np.arange(n, dtype=np.int64) * mult) % n
        orig_of = np.empty(n, np.int64)
        orig_of[pos_of] = np.arange(n)
        return orig_of, pos_of

    # The library indexes one representative per group of identical rows (sg_collapse.hip): the ranges of the self-join
    # form are then ranges of GROUPS, the ranks' blocks hold groups, and the gathered result is expanded to rows
    # (sg_postings_rows, sg_topn_expand_groups).  ``grouped = True`` restates that contract.
    grouped = False

    def postings(self, m, tile_cols=0, permute=True):
        if not self.grouped:
            return m
        return GroupedIndex(m)

    def index_is_permuted(self, post):
        return self.permuted

    def selfjoin_rows(self, A_full, post):
        return post.unique.shape[0] if isinstance(post, GroupedIndex) else A_full.shape[0]

    @staticmethod
    def expand_rows(post, rows, cols, vals, counts):
        """sg_topn_expand_groups restated: result rows of the caller's ``rows`` from the result over groups."""
        stride = cols.shape[1]
        out_c = np.zeros((len(rows), stride), np.int32)
        out_v = np.zeros((len(rows), stride), vals.dtype)
        out_n = np.zeros(len(rows), np.int32)
        for k, i in enumerate(rows):
            g = post.gid[i]
            c, v = [], []
            for e in range(counts[g]):
                mem = post.members[cols[g, e]]
                c.extend(mem)
                v.extend([vals[g, e]] * len(mem))
            c, v = np.asarray(c, np.int64), np.asarray(v, vals.dtype)
            order = np.lexsort((c, -v))[:stride]
            out_n[k] = len(order)
            out_c[k, :len(order)] = c[order]
            out_v[k, :len(order)] = v[order]
        return out_c, out_v, out_n

    def multiply(self, left, right, top_n, threshold):
        if isinstance(right, GroupedIndex):
            right = right.full
        C = P.sp_matmul_topn_port(left, right.T, top_n, threshold, True, 2)
        stride = max(1, min(top_n, right.shape[0]))
        n = left.shape[0]
        cols = np.zeros((n, stride), np.int32)
        vals = np.zeros((n, stride), self.dtype)
        cnt = np.diff(C.indptr).astype(np.int32)
        for i in range(n):
            lo, hi = C.indptr[i], C.indptr[i + 1]
            cols[i, :hi - lo] = C.indices[lo:hi]
            vals[i, :hi - lo] = C.data[lo:hi]
        return cols, vals, cnt

    def keep_alive(self, res, *objs):
        pass

    # ---- the self-join form over row ranges: the contract of sg_selfjoin_range / sg_selfjoin_merge restated with the
    #      oracle's multiply (rows of the range keep their matches j <= i; mirrored pairs (i, j < i, score) go out)
    def selfjoin_range(self, A_full, post, top_n, threshold, lo, hi, step=1):
        from string_grouper_amd.distributed import share_positions
        if isinstance(post, GroupedIndex):
            # the result over GROUPS must hold what expansion needs: a row's top_n columns can come from top_n groups at
            # most, so top_n groups per group are enough (the library's argument, sg_collapse.hip)
            part = self.selfjoin_range(post.unique, post.unique, top_n, threshold, lo, hi, step)
            part["groups_of"] = post
            return part
        n = A_full.shape[0]
        orig_of, pos_of = self.permutation(n) if self.permuted else (np.arange(n), np.arange(n))
        rows = orig_of[share_positions(lo, hi, step)]       # the rows of the share (of positions)
        C = P.sp_matmul_topn_port(A_full[rows], A_full.T, n, threshold, True, 2)      # every match of the rows
        stride = max(1, min(top_n, n))
        cols = np.zeros((n, stride), np.int32)
        vals = np.zeros((n, stride), self.dtype)
        cnt = np.zeros(n, np.int32)
        pairs = []
        for r in range(len(rows)):
            i = int(rows[r])
            a, b = C.indptr[r], C.indptr[r + 1]
            j, sc = C.indices[a:b], C.data[a:b]
            own = pos_of[j] <= pos_of[i]                    # port order: score descending, column ascending
            k = min(int(own.sum()), stride)
            cols[i, :k] = j[own][:k]
            vals[i, :k] = sc[own][:k]
            cnt[i] = k
            for jj, ss in zip(j[pos_of[j] < pos_of[i]], sc[pos_of[j] < pos_of[i]]):
                pairs.append((i, int(jj), ss))
        words = 4 if np.dtype(self.dtype) == np.float64 else 3
        flat = np.zeros((len(pairs), words), np.int32)
        for p, (i, jj, ss) in enumerate(pairs):
            flat[p, 0], flat[p, 1] = i, jj
            flat[p, 2:] = np.frombuffer(np.asarray([ss], self.dtype).tobytes(), np.int32)
        return {"res": (cols, vals, cnt), "pairs": torch.from_numpy(flat.reshape(-1)), "words": words, "top_n": stride}

    def selfjoin_pairs(self, part):
        return part["pairs"]

    def selfjoin_discard(self, part):
        pass

    def selfjoin_merge(self, part, pairs_all, lo, hi, step=1):
        from string_grouper_amd.distributed import share_positions
        cols, vals, cnt = part["res"]
        words, stride = part["words"], part["top_n"]
        rec = pairs_all.numpy().reshape(-1, words)
        scores = np.frombuffer(np.ascontiguousarray(rec[:, 2:]).tobytes(), self.dtype)
        n = cols.shape[0]
        orig_of = self.permutation(n)[0] if self.permuted else np.arange(n)
        my_rows = orig_of[share_positions(lo, hi, step)]
        for row in my_rows:
            mine = np.flatnonzero(rec[:, 1] == row)
            if len(mine) == 0:
                continue
            c = np.concatenate([cols[row, :cnt[row]], rec[mine, 0]])
            v = np.concatenate([vals[row, :cnt[row]], scores[mine]])
            order = np.lexsort((c, -v))[:stride]
            cnt[row] = len(order)
            cols[row, :len(order)] = c[order]
            vals[row, :len(order)] = v[order]
        post = part.get("groups_of")
        if post is not None:
            # the range was one of groups: the rank's rows are the members of its groups, expanded from the tables of the
            # index (which every rank holds)
            mine = set(int(g) for g in my_rows)
            rows = np.array([i for i in range(post.full.shape[0]) if int(post.gid[i]) in mine], np.int64)
            return PermutedBlock(self.expand_rows(post, rows, cols, vals, cnt), None, torch.from_numpy(rows.astype(np.int32)))
        if step > 1:     # an interleaved share: the block lists its rows
            return PermutedBlock((cols[my_rows], vals[my_rows], cnt[my_rows]), None, torch.from_numpy(my_rows.astype(np.int64)))
        if self.permuted:
            ids = orig_of[lo:hi]
            return PermutedBlock((cols[ids], vals[ids], cnt[ids]), torch.from_numpy(orig_of))
        return cols[lo:hi], vals[lo:hi], cnt[lo:hi]

    def topn_tensors(self, res):
        if isinstance(res, PermutedBlock):
            res = res.arrays
        return torch.from_numpy(res[0]), torch.from_numpy(res[1]), torch.from_numpy(res[2])


class PermutedBlock:
    """A rank's block of the self-join form when the ranges are ranges of positions: rows orig_of[lo:hi] in that order
    (what distributed.TopNRows is for the device library)."""

    def __init__(self, arrays, orig_of, row_ids=None):
        self.arrays, self.orig_of, self.row_ids = arrays, orig_of, row_ids

    def __getitem__(self, k):
        return self.arrays[k]


class GroupedIndex:
    """The index over one representative per group of identical rows: groups numbered by ascending lowest member."""

    def __init__(self, m):
        m = sp.csr_matrix(m)
        self.full = m
        key_to_group, self.members, self.gid = {}, [], np.zeros(m.shape[0], np.int64)
        for i in range(m.shape[0]):
            a, b = m.indptr[i], m.indptr[i + 1]
            key = (m.indices[a:b].tobytes(), m.data[a:b].tobytes())
            g = key_to_group.setdefault(key, len(self.members))
            if g == len(self.members):
                self.members.append([])
            self.members[g].append(i)
            self.gid[i] = g
        self.unique = m[[mem[0] for mem in self.members]]
