"""Ties the committed counter passes (profiles/k4_traffic.json, profiles/k4_counters.json) to the kernel source they were
measured on.  The passes cannot run inside bench.py's timed run (rocprofv3 --pmc is a run of its own), so the bench line
quotes committed files; `source_sha` -- written by scripts/pmc_traffic.py / pmc_counters.py when they make those files --
is the SHA-256 of the sources of the multiply's kernels at that moment, and bench.py drops the quoted fields (and says
"stale") when the sources have changed since."""
from __future__ import annotations

import hashlib
import json
import os

# the files whose code decides what the dominant kernel fetches and issues (K3 lays the index out, K4p reads it)
KERNEL_SOURCES = ("string_grouper_amd/csrc/sg_spgemm_pruned.hip", "string_grouper_amd/csrc/sg_postings.hip",
                  "string_grouper_amd/csrc/sg_internal.h", "string_grouper_amd/csrc/sg_k4_device.h")


def kernel_source_sha(root: str) -> str:
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(root, rel), "rb") as f:
            h.update(rel.encode() + b"\0" + f.read() + b"\0")
    return h.hexdigest()


def committed_counters(root: str, rows: int, dtype: str, kernel_tag: str, kernel_ms: float) -> dict:
    """The roofline fields bench.py takes from the committed passes, for the workload (rows, dtype) and kernel it has just
    timed: traffic (+ source, note) and valu_issue_frac (+ instructions, source) -- or, for a file whose `source_sha` is
    not the current sources', `<field>_stale` with both hashes and no number."""
    out: dict = {}
    sha = kernel_source_sha(root)

    def load(prefix):
        """The committed pass of this workload: profiles/<prefix>.json (the headline's) or profiles/<prefix>_<tag>.json
        (other sizes: k4_traffic_5M.json ...), whichever names (rows, dtype, kernel)."""
        import glob
        for path in sorted(glob.glob(os.path.join(root, "profiles", prefix + "*.json"))):
            try:
                with open(path) as f:
                    d = json.load(f)
            except Exception:
                continue
            if d.get("workload_rows") == rows and d.get("dtype") == dtype and d.get("kernel", "K4") == kernel_tag:
                d["_file"] = os.path.basename(path)
                return d
        return None

    tr = load("k4_traffic")
    if tr is not None:
        if tr.get("source_sha") == sha:
            out["traffic"] = tr["traffic_bytes_per_launch_raw"]
            out["traffic_source"] = (f"committed PMC pass (profiles/{tr['_file']}, written by scripts/pmc_traffic.py on "
                                     f"kernel sources {sha[:12]}); not measured in this run")
            out["traffic_note"] = tr["source"] + "; " + tr["note"]
            if isinstance(tr.get("tcc"), dict) and "hit_frac" in tr["tcc"]:      # (an L2 pass of the same session, where one was made)
                out["tcc_hit_frac"] = tr["tcc"]["hit_frac"]
        else:
            out["traffic"] = None
            out["traffic_stale"] = {"status": "stale", "measured_on_sources": (tr.get("source_sha") or "unrecorded")[:12],
                                    "current_sources": sha[:12]}
    kc = load("k4_counters")
    if kc is not None:
        if kc.get("source_sha") == sha:
            valu = float(kc["per_launch"]["SQ_INSTS_VALU"])
            out["valu_issue_frac"] = valu * 4.0 / 1024.0 / 2.4e9 / (kernel_ms * 1e-3)
            out["valu_insts_per_launch"] = valu
            out["valu_source"] = (f"committed PMC pass (profiles/{kc['_file']}, written by scripts/pmc_counters.py on kernel "
                                  f"sources {sha[:12]}); instruction counts do not depend on the run, the kernel time is this run's")
        else:
            out["valu_stale"] = {"status": "stale", "measured_on_sources": (kc.get("source_sha") or "unrecorded")[:12],
                                 "current_sources": sha[:12]}
    return out
