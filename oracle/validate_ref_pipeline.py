"""Shows that oracle/ref_pipeline.py restates the reference's match_strings: the same frames, the same time.
Needs /root/reference (this container, not the GPU box).  Run:  python -m oracle.validate_ref_pipeline [n]
The log is committed under profiles/."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["SG_SHIM_BACKEND"] = "port"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "ref_shims"))
sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402
import pandas as pd  # noqa: E402
import string_grouper as ref  # noqa: E402

assert "/root/reference" in ref.__file__
from oracle import ref_pipeline as R  # noqa: E402
from string_grouper_amd.synth import synth_names  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 4
names = pd.Series(synth_names(n, 1234), name="name")
kw = dict(max_n_matches=10, min_similarity=0.8, tfidf_matrix_dtype=np.float32, number_of_processes=threads)
print(f"{n} names, self-join, ntop 10, min_sim 0.8, fp32, number_of_processes={threads}, "
      f"cores available {len(os.sched_getaffinity(0))}", flush=True)
for rep in range(2):
    t0 = time.perf_counter()
    want = ref.match_strings(names, **kw)
    t_ref = time.perf_counter() - t0
    tm = {}
    t0 = time.perf_counter()
    got = R.match_strings_cpu(names, timings=tm, **kw)
    t_res = time.perf_counter() - t0
    same = list(got.columns) == list(want.columns) and len(got) == len(want) and all(
        np.array_equal(got[c].to_numpy(), want[c].to_numpy()) for c in want.columns)
    print(f"run {rep}: unmodified reference {t_ref:7.2f} s   restatement {t_res:7.2f} s   ratio {t_res / t_ref:5.3f}   "
          f"frames identical: {same}   rows {len(want)}", flush=True)
    print("   restatement split: " + ", ".join(f"{k} {v:.2f}" for k, v in tm.items()), flush=True)
    assert same
