"""Summarise a rocprofv3 --pmc output directory: per kernel name, mean counter value per dispatch."""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
if not files:
    print("no counter_collection.csv under", d)
    sys.exit(0)
acc = defaultdict(lambda: defaultdict(list))
for f in files:
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row.get("Kernel_Name", "?")[:60]
            acc[name][row.get("Counter_Name", "?")].append(float(row.get("Counter_Value", 0)))
for name, ctrs in sorted(acc.items(), key=lambda kv: -sum(sum(v) for v in kv[1].values()))[:6]:
    print(name)
    for c, vals in sorted(ctrs.items()):
        print(f"   {c:28s} n={len(vals):4d} mean={sum(vals) / len(vals):.6g} max={max(vals):.6g}")
