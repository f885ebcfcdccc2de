"""The timed CPU baseline of bench.py (``cpu_baseline``) -- TEST INFRASTRUCTURE, never on the product path.

Runs as a CHILD PROCESS of bench.py pinned to the first ``--cores`` CPUs it is allowed to use (the reference's
README quotes its numbers at ``number_of_processes=4``, README.md:42-56; BASELINE.json's north star names "the
reference 4-core CPU match_strings wall-clock"), on a BOUNDED sample of the benchmark workload, and prints one
JSON object.  What is timed is oracle/ref_pipeline.py -- the reference's match_strings call sequence restated on
sklearn + the C port of sparse_dot_topn (the reference package is Python under /root/reference and does not
exist on the GPU box; oracle/validate_ref_pipeline.py shows restatement == unmodified reference in output and time).

Composition of the full-size estimate (every term reported):
  * vectorise : the three tokenisation passes of the reference (ctor fit, fit, transform), single-threaded Python
                + sklearn, timed in full on ``n_small`` names and scaled by rows (the per-string cost does not
                depend on the list length);
  * multiply  : the blocked product exactly as the reference would cut the FULL problem (n_blocks from its own
                guess, string_grouper.py:387-389): every right-hand block is converted and multiplied, for the
                first S left rows only; the scan part is scaled by the exact count of intermediate products
                (MACs), the per-block CSC->CSR conversions are not scaled (the full run pays them once, too);
  * tail      : lil round trip (diagonal, symmetrise), match list, frames -- timed in full at ``n_small``, scaled
                by the number of match rows.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_quota():
    """CPUs the cgroup grants this process (None: unlimited / unknown) -- the affinity mask alone can overstate it."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, period = f.read().split()
        if q != "max":
            return max(1, int(int(q) / int(period)))
    except (OSError, ValueError):
        pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--matrix", required=True, help="npz with the full TF-IDF CSR (indptr, indices, data, shape)")
    ap.add_argument("--rows", type=int, required=True)
    ap.add_argument("--top-n", type=int, default=10)
    ap.add_argument("--min-similarity", type=float, default=0.8)
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--cores", type=int, default=4)
    ap.add_argument("--n-small", type=int, default=40000)
    ap.add_argument("--multiply-seconds", type=float, default=10.0)
    ap.add_argument("--matches-full", type=int, default=0, help="match rows of the full job (from the GPU run)")
    ap.add_argument("--macs-full", type=int, default=0, help="intermediate products of the full job")
    ap.add_argument("--multiply-only", action="store_true", help="skip the vectorise / tail legs (all-cores line)")
    args = ap.parse_args()

    allowed = sorted(os.sched_getaffinity(0))
    quota = cpu_quota()
    cores = max(1, min(args.cores, len(allowed), quota or len(allowed)))   # never more threads than the cgroup grants
    os.sched_setaffinity(0, set(allowed[:cores]))
    os.environ["OMP_NUM_THREADS"] = str(cores)
    os.environ["OMP_WAIT_POLICY"] = "passive"      # 166 short parallel regions per pass: no spinning at their barriers

    import numpy as np
    import pandas as pd
    import scipy.sparse as sp

    from oracle import oracle as O
    from oracle import port as P
    from oracle import ref_pipeline as R
    from string_grouper_amd.synth import synth_names

    dtype = np.float32 if args.dtype == "f32" else np.float64
    z = np.load(args.matrix)
    A = sp.csr_matrix((z["data"], z["indices"], z["indptr"]), shape=tuple(z["shape"]))
    n = A.shape[0]
    out = {"cores": cores, "cpu_model": cpu_model(), "kind": "port", "rows": n, "cpus_in_affinity_mask": len(allowed),
           "cgroup_cpu_quota": quota}

    # ---- multiply: the reference's own block split of the full problem, first S left rows
    n_blocks = O.guess_n_blocks(n, n)
    chunks = O.define_chunks(n, n_blocks[1])
    t0 = time.perf_counter()
    Bs = [A[c.start:c.stop] for c in chunks]
    BTs = [P._as_bt_csr(Bi.T) for Bi in Bs]              # what every sp_matmul_topn call does to its right operand
    t_convert = time.perf_counter() - t0
    df = np.bincount(A.indices, minlength=A.shape[1]).astype(np.int64)
    macs_row = np.add.reduceat(df[A.indices], A.indptr[:-1].astype(np.int64))
    macs_row[np.diff(A.indptr) == 0] = 0
    macs_full = int(macs_row.sum())

    def products(S):
        t0 = time.perf_counter()
        Cs = [P.sp_matmul_topn_port(A[:S], Bi.T, args.top_n, args.min_similarity, True, cores) for Bi in Bs]
        t1 = time.perf_counter()
        R.zip_port(args.top_n, Cs)
        return t1 - t0, time.perf_counter() - t1

    cum_macs = np.cumsum(macs_row)
    S = min(n, 2000)
    for _ in range(3):                                  # grow the sample until it fills the time budget
        t_prod, t_zip = products(S)
        t_scan = max(t_prod - t_convert, 1e-3)
        if S >= n or t_scan >= 0.5 * args.multiply_seconds:
            break
        want_macs = cum_macs[S - 1] * args.multiply_seconds / t_scan
        S_next = int(min(n, max(S + 1, np.searchsorted(cum_macs, want_macs))))
        if S_next <= S:
            break
        S = S_next
    macs_sample = int(macs_row[:S].sum())
    t_scan = max(t_prod - t_convert, 0.0)
    mult_full = t_convert + t_scan * macs_full / max(macs_sample, 1) + t_zip * n / S
    out["multiply"] = {"n_blocks": list(n_blocks), "sample_left_rows": S, "seconds_sample": t_prod + t_zip,
                       "seconds_block_conversions": t_convert, "macs_sample": macs_sample, "macs_full": macs_full,
                       "seconds_full_estimate": mult_full, "rows_per_s": n / mult_full}
    if args.multiply_only:
        print(json.dumps(out))
        return

    # ---- the whole reference pipeline at n_small (vectorise passes and the tail are timed here)
    n_small = min(args.n_small, n)
    names = pd.Series(synth_names(n_small, 1234), name="name")
    tm = {}
    t0 = time.perf_counter()
    frame = R.match_strings_cpu(names, max_n_matches=args.top_n, min_similarity=args.min_similarity,
                                tfidf_matrix_dtype=dtype, number_of_processes=cores, timings=tm)
    t_small = time.perf_counter() - t0
    # three passes at the steady-state rate (the first one also pays one-off imports, which do not scale with rows)
    t_vec = 1.5 * (tm["vectorise_pass2_fit"] + tm["vectorise_pass3_transform"])
    t_tail = tm.get("lil_diagonal_symmetrise", 0.0) + tm["matches_list"] + tm["get_matches_frames"] + tm["vstack"]
    matches_full = args.matches_full or int(len(frame) * n / n_small)
    vec_full = t_vec * n / n_small
    tail_full = t_tail * matches_full / max(len(frame), 1)
    total = vec_full + mult_full + tail_full
    out["small_run"] = {"rows": n_small, "seconds": t_small, "match_rows": int(len(frame)), "split": tm}
    out["vectorise"] = {"seconds_full_estimate": vec_full, "rows_per_s_per_pass": 3.0 * n_small / t_vec, "threads": 1}
    out["tail"] = {"seconds_full_estimate": tail_full, "match_rows_full": matches_full}
    out["seconds_full_estimate"] = total
    out["value"] = n / total
    out["unit"] = "rows/s"
    print(json.dumps(out))


if __name__ == "__main__":
    main()
