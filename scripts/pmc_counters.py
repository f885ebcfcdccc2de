"""Instruction counters of the dominant multiply kernel from a rocprofv3 --pmc pass (SQ_INSTS_VALU ... : pass 1 of
scripts/gpu_session.sh pmcsq) -> a JSON that bench.py reads for roofline.valu_issue_frac (profiles/k4_counters.json).

    python scripts/pmc_counters.py <pmc output dir> <out.json> [rows=663000] [dtype=f32]
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

d, out = sys.argv[1], sys.argv[2]
rows = int(sys.argv[3]) if len(sys.argv) > 3 else 663000
dtype = sys.argv[4] if len(sys.argv) > 4 else "f32"
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            acc[row.get("Kernel_Name", "?")][row.get("Counter_Name", "?")].append(float(row.get("Counter_Value", 0)))
# the kernel with the most VALU instructions in all: the dominant multiply kernel
best = max(acc.items(), key=lambda kv: sum(kv[1].get("SQ_INSTS_VALU", [0])), default=None)
if best is None or not best[1].get("SQ_INSTS_VALU"):
    print("no SQ_INSTS_VALU rows under", d)
    sys.exit(1)
name, ctrs = best
mean = {c: sum(v) / len(v) for c, v in ctrs.items()}
sym = ", true, " in name.split("(")[0]
kind = ("K4p-sym" if sym else "K4p") if "pruned" in name else "K4"
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from string_grouper_amd._provenance import kernel_source_sha  # noqa: E402
json.dump({"workload_rows": rows, "dtype": dtype, "kernel": kind, "source_sha": kernel_source_sha(ROOT), "kernel_name": name.split("(")[0],
           "per_launch": {c: mean[c] for c in sorted(mean)}, "launches_averaged": len(ctrs["SQ_INSTS_VALU"]),
           "source": "scripts/pmc_counters.py over one rocprofv3 --pmc pass (SQ_* counters only, no trace domains) of "
                     "`python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-exact-kernel --no-end-to-end --no-side-runs`",
           "note": "wave-level instruction counts; VALU issue fraction = SQ_INSTS_VALU x 4 cycles / 1024 SIMDs / 2.4 GHz / kernel time"},
          open(out, "w"), indent=1)
print(kind, {c: f"{v:.4g}" for c, v in sorted(mean.items())})
