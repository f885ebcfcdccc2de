"""Print a rocprofv3 kernel_stats.csv compactly: python scripts/kstats.py <csv> [rows=16]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for row in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 16]:
    n = row["Name"]
    m = re.search(r"(\w+)(<[^>]*>)?\(", n)
    name = (m.group(1) + (m.group(2) or "")) if m else n[:60]
    print(f"{name:64s} calls {row['Calls']:>4s} avg {float(row['AverageNs']) / 1e6:8.3f} ms  total {float(row['TotalDurationNs']) / 1e6:8.2f} ms")
