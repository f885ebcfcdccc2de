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


ARCHIVE = os.path.join(ROOT, "oracle", "_ref", "reference_pkg.zip")


def load_reference():
    """The UNMODIFIED reference package, imported from the build output oracle/_ref/reference_pkg.zip (the packed
    /root/reference/string_grouper; oracle/mount_reference.py) with the one third-party dependency this image lacks --
    the sparse_dot_topn wheel -- stood in for by the C port (tests/ref_shims, SG_SHIM_BACKEND=port).  Returns (module
    string_grouper.string_grouper, directory to remove afterwards) or (None, None) when the archive is not there."""
    import tempfile
    import zipfile
    if not os.path.exists(ARCHIVE):
        return None, None
    d = tempfile.mkdtemp(prefix="sg_ref_baseline_")
    with zipfile.ZipFile(ARCHIVE) as z:
        z.extractall(d)
    os.environ["SG_SHIM_BACKEND"] = "port"
    # in front of the repository root, which holds a drop-in alias package of the same name
    sys.path.insert(0, d)
    sys.path.insert(0, os.path.join(ROOT, "tests", "ref_shims"))
    import string_grouper.string_grouper as ref
    assert ref.__file__.startswith(d), ref.__file__
    return ref, d


def time_reference(ref, names, top_n, min_similarity, dtype, cores):
    """``ref.match_strings(names, ...)`` -- string_grouper.py:130-153, unmodified -- once, on the wall clock, with the time
    spent inside its legs read off by wrapping (not editing) the functions it calls: the constructor's fit (:305-308),
    _get_tf_idf_matrices (:685-697: the second fit and the transform), the sparse_dot_topn calls of _build_matches
    (:725-746).  What is left is the tail: vstack, lil diagonal + symmetrise, match list, frames."""
    legs = {"vectorise_ctor_fit": 0.0, "vectorise_fit_transform": 0.0, "multiply_calls": 0.0}
    saved = (ref.StringGrouper._build_corpus, ref.StringGrouper._get_tf_idf_matrices, ref.sp_matmul_topn, ref.zip_sp_matmul_topn)

    def timed(fn, key):
        def wrapper(*a, **k):
            t = time.perf_counter()
            try:
                return fn(*a, **k)
            finally:
                legs[key] += time.perf_counter() - t
        return wrapper

    ref.StringGrouper._build_corpus = timed(saved[0], "vectorise_ctor_fit")
    ref.StringGrouper._get_tf_idf_matrices = timed(saved[1], "vectorise_fit_transform")
    ref.sp_matmul_topn = timed(saved[2], "multiply_calls")
    ref.zip_sp_matmul_topn = timed(saved[3], "multiply_calls")
    try:
        t0 = time.perf_counter()
        frame = ref.match_strings(names, max_n_matches=top_n, min_similarity=min_similarity, tfidf_matrix_dtype=dtype,
                                  number_of_processes=cores)
        total = time.perf_counter() - t0
    finally:
        (ref.StringGrouper._build_corpus, ref.StringGrouper._get_tf_idf_matrices, ref.sp_matmul_topn,
         ref.zip_sp_matmul_topn) = saved
    legs["tail"] = max(total - sum(legs.values()), 0.0)
    return total, legs, frame


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
    ap.add_argument("--reference-full", action="store_true",
                    help="also run the unmodified reference's match_strings on ALL rows (minutes; bench.py --cpu-full)")
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
        if args.multiply_seconds < 1000:                # a bounded sample stays one: the first estimate (fixed costs in its
            S_next = min(S_next, 12 * S)                #  denominator) overshoots by an order of magnitude
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

    # ---- the whole pipeline at n_small: vectorise passes and tail.  The UNMODIFIED reference when its package travelled
    # (oracle/_ref/reference_pkg.zip: `kind` "reference+port" -- the reference's own match_strings, its one absent native
    # dependency stood in for by the C port); the restated call sequence (oracle/ref_pipeline.py) beside it as a cross-check,
    # and alone ("port") where the archive is missing.
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
    restated = {"rows": n_small, "seconds": t_small, "match_rows": int(len(frame)), "split": tm,
                "vectorise_seconds": t_vec, "tail_seconds": t_tail}
    ref, ref_dir = load_reference()
    if ref is not None:
        try:
            time_reference(ref, names[:2000], args.top_n, args.min_similarity, dtype, cores)        # (imports, first-call costs)
            r_total, r_legs, r_frame = time_reference(ref, names, args.top_n, args.min_similarity, dtype, cores)
            same = bool(len(r_frame) == len(frame) and
                        np.array_equal(r_frame["similarity"].to_numpy(), frame["similarity"].to_numpy()) and
                        np.array_equal(r_frame.iloc[:, 0].to_numpy(), frame.iloc[:, 0].to_numpy()))
            out["kind"] = "reference+port"
            out["reference_run"] = {"rows": n_small, "seconds": r_total, "match_rows": int(len(r_frame)), "legs": r_legs,
                                    "what": "string_grouper.match_strings of the unmodified reference package (packed from "
                                            "/root/reference by oracle/mount_reference.py), sparse_dot_topn = oracle/sdtn_port.c",
                                    "frames_equal_the_restated_pipeline": same}
            t_vec = r_legs["vectorise_ctor_fit"] + r_legs["vectorise_fit_transform"]
            t_tail = r_legs["tail"]
            frame = r_frame
            if args.reference_full:
                full_names = pd.Series(synth_names(n, 1234), name="name")
                f_total, f_legs, f_frame = time_reference(ref, full_names, args.top_n, args.min_similarity, dtype, cores)
                out["reference_full_run"] = {"rows": n, "seconds": f_total, "match_rows": int(len(f_frame)), "legs": f_legs}
        finally:
            import shutil
            shutil.rmtree(ref_dir, ignore_errors=True)
    matches_full = args.matches_full or int(len(frame) * n / n_small)
    vec_full = t_vec * n / n_small
    tail_full = t_tail * matches_full / max(len(frame), 1)
    total = vec_full + mult_full + tail_full
    out["small_run"] = restated
    out["vectorise"] = {"seconds_full_estimate": vec_full, "rows_per_s_per_pass": 3.0 * n_small / t_vec, "threads": 1,
                        "scaled_from_rows": n_small}
    out["tail"] = {"seconds_full_estimate": tail_full, "match_rows_full": matches_full, "scaled_from_rows": n_small}
    out["seconds_full_estimate"] = total
    out["value"] = n / total
    if "reference_full_run" in out:          # measured, not composed
        out["seconds_full_estimate"] = out["reference_full_run"]["seconds"]
        out["value"] = n / out["reference_full_run"]["seconds"]
    out["unit"] = "rows/s"
    print(json.dumps(out))


if __name__ == "__main__":
    main()
