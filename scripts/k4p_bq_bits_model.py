"""CPU model: how many more pairs would the pruned multiply score exactly if the filter postings carried their value in
8 bits (a byte the multiply could select with SDWA, saving one `v_and` per slot) instead of the 11 bits a 4096-column tile
leaves?  The kernel's integer filter arithmetic (tests/test_prune_model.py) on a sample of rows of the 663 k workload.

    python scripts/k4p_bq_bits_model.py [rows=663000] [sample=150]

Result at 663 k (120 rows): 8 bits +2 %, 6 bits +7 % pairs.  The precision would be affordable -- but the instruction is not
saved: `v_mul_hi_u32_u24` wants the value IN PLACE in bits 16..23 so that its >> 32 does the scaling, and SDWA's BYTE_2
select delivers it in bits 0..7 (product < 2^32, high half zero); `v_mul_u32_u24` + a shift is the same two operations as
`v_and` + `v_mul_hi`.  Idea closed.
"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from oracle import oracle as O  # noqa: E402
from string_grouper_amd.synth import synth_names  # noqa: E402

f32 = np.float32
AB = 13


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 663000
    n_sample = int(sys.argv[2]) if len(sys.argv) > 2 else 150
    thr, delta = 0.8, 0.05
    t0 = time.time()
    names = synth_names(n, 1234)
    (m,), _, _ = O.tfidf_sklearn(names, [names], dtype=np.float32)
    m = m.tocsr()
    m.sort_indices()
    mt = m.T.tocsr()
    mt.sort_indices()
    print(f"# {m.shape}, nnz {m.nnz}, {time.time() - t0:.0f} s", flush=True)
    df_all = np.diff(mt.indptr).astype(np.int64)
    freq_min = max(1, int(0.0045 * n))
    norm_up = np.nextafter(f32(np.sqrt(f32(np.asarray(m.multiply(m).sum(axis=1)).max())) * f32(1.000001)), f32(2))
    inv = f32(1.0) / f32(norm_up)
    rows_all = np.repeat(np.arange(n), np.diff(m.indptr))
    frequent = df_all >= freq_min
    f2 = np.bincount(rows_all, weights=(m.data.astype(np.float64) ** 2) * frequent[m.indices], minlength=n)
    fq = np.minimum(255, np.ceil(np.nextafter(np.sqrt(f2).astype(f32), f32(2)) * inv * f32(255.0) * f32(1.000002))).astype(np.int64)
    rng = np.random.default_rng(11)
    sample = np.sort(rng.choice(n, n_sample, replace=False))
    out, alt = {}, {}
    for bits in (11, 8, 6):
        bq_max = (1 << bits) - 1
        bq_t = np.minimum(bq_max, np.ceil(mt.data.astype(f32) * inv * f32(bq_max) * f32(1.000002))).astype(np.int64)
        surv_total = 0
        for i in sample:
            lo, hi = m.indptr[i], m.indptr[i + 1]
            k, a = m.indices[lo:hi], m.data[lo:hi]
            nnz = len(k)
            if nnz == 0 or nnz > 64:
                continue
            df = df_all[k]
            w = (a.astype(f32) * a.astype(f32) * f32(1.00001)).astype(np.float64)
            order = np.lexsort((np.arange(nnz), -df))
            cum = np.empty(nnz)
            cum[order] = np.cumsum(w[order])
            beta = thr - delta
            budget = (beta / float(norm_up)) ** 2 * (1.0 - 1e-6)
            in_s = (cum <= budget) & (df >= freq_min)
            in_p = ~in_s
            if not in_p.any():
                continue
            bs2 = cum[in_s].max() if in_s.any() else 0.0
            b_s = f32(f32(np.sqrt(f32(bs2))) * f32(1.000002))
            t0_ = f32(f32(f32(thr) - f32(1e-5)) * f32(32768.0)) - f32(2.0)
            c1 = f32(f32(f32(b_s * f32(norm_up)) * f32(32768.0 / 255.0)) * f32(1.000002))
            n_p = int(in_p.sum())
            T0 = int(np.floor(f32(t0_ * f32(256.0)))) - 256 * n_p
            C1 = int(f32(c1 * f32(256.0))) + 1
            q = np.zeros(n, np.int64)
            for t in np.nonzero(in_p)[0]:
                sl = slice(mt.indptr[k[t]], mt.indptr[k[t] + 1])
                c_a = f32(f32(f32(f32(a[t]) * f32(norm_up)) * f32(32768.0 / bq_max)) * f32(1.000002))
                # CA scaled so that (CA * bq) >> SH is an upper bound of a * b * 2^15 - 1 (the kernel: 24-bit multiplies)
                CA = int(f32(c_a * f32(1 << 16))) + 1
                x = (CA * bq_t[sl]) >> 16
                cols = mt.indices[sl]
                keep = cols <= i
                q[cols[keep]] += x[keep]
            touched = np.flatnonzero(q)
            tq = (T0 - C1 * fq[touched]) >> 8
            surv_total += int((q[touched] >= tq).sum())
            if bits == 11:
                # the per-column threshold with an INTEGER slope (no >> 8 per slot): tq' = (T0 >> 8) - ceil(C1 / 256) * fq <= tq
                tq_i = (T0 >> 8) - (-(-C1 // 256)) * fq[touched] - 1
                alt["integer slope"] = alt.get("integer slope", 0) + int((q[touched] >= tq_i).sum())
        out[bits] = surv_total / n_sample
        print(f"value field of {bits} bits: {out[bits]:.1f} pairs scored exactly per row ({100 * out[bits] / out[11]:.0f} %)", flush=True)
        if bits == 11:
            for k_, v in alt.items():
                print(f"  {k_}: {v / n_sample:.1f} ({100 * v / n_sample / out[11]:.0f} %)", flush=True)


if __name__ == "__main__":
    main()
