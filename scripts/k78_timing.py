"""Timing of the reductions over the match list: device (K7 / K8) vs the host formulation.  Development tool."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pandas as pd
import string_grouper_amd as sga
import string_grouper_amd.engine as E
from string_grouper_amd import _native as N
from string_grouper_amd.synth import synth_names

ctx = N.default_context(0)
E.set_engine(E.HipEngine(ctx))


def timed(f):
    t0 = time.perf_counter()
    r = f()
    return time.perf_counter() - t0, r


n = int(sys.argv[1]) if len(sys.argv) > 1 else 663000
s = pd.Series(synth_names(n, 1234))
for rep in ("centroid", "first"):
    t_fit, sg = timed(lambda: sga.StringGrouper(s, min_similarity=0.8, tfidf_matrix_dtype=np.float32, group_rep=rep,
                                                ignore_index=True).fit())
    t_dev, g_dev = timed(sg.get_groups)
    t_red_dev, _ = timed(lambda: sg.__dict__["_device_matches"].group_reps(rep == "centroid"))
    sg._drop_device_matches()
    t_host, g_host = timed(sg.get_groups)
    t_red_host, _ = timed(lambda: sg._group_reps_on_host(n))
    print(json.dumps({"what": "group_similar_strings", "n": n, "group_rep": rep, "fit_s": t_fit, "match_list_rows": len(sg._matches_list),
                      "get_groups_device_s": t_dev, "get_groups_host_s": t_host, "reduction_device_s": t_red_dev,
                      "reduction_host_s": t_red_host, "identical": bool(g_dev.equals(g_host))}), flush=True)

n_m, n_d = n, n // 3
master = synth_names(n_m, 1234)
dupes = synth_names(n_d, seed=4321, perturb_of=master, perturb_frac=0.5)
m, d = pd.Series(master), pd.Series(dupes)
t_fit, sg = timed(lambda: sga.StringGrouper(m, d, min_similarity=0.7, max_n_matches=20, tfidf_matrix_dtype=np.float32,
                                            ignore_index=True).fit())
t_dev, g_dev = timed(sg.get_groups)
t_red_dev, _ = timed(lambda: sg.__dict__["_device_matches"].best_master())
sg._drop_device_matches()
t_host, g_host = timed(sg.get_groups)
print(json.dumps({"what": "match_most_similar", "n_master": n_m, "n_duplicates": n_d, "fit_s": t_fit,
                  "match_list_rows": len(sg._matches_list), "get_groups_device_s": t_dev, "get_groups_host_s": t_host,
                  "reduction_device_s": t_red_dev, "identical": bool(g_dev.equals(g_host))}), flush=True)
