"""Idle gaps of the GPU inside one step of the hot path, from a rocprofv3 kernel trace:
python scripts/ktrace_gaps.py <kernel_trace.csv> [min_gap_us=8]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda e: e[0])


def short(n):
    m = re.search(r"(\w+)(<[^>]*>)?\(", n)
    return (m.group(1) if m else n)[:40]


# the last step: from the last tokenize_short_kernel on
starts = [i for i, e in enumerate(ev) if "tokenize_short_kernel" in e[2]]
i0 = starts[-1]
seq = ev[i0:]
t0 = seq[0][0]
busy = 0
prev_end = seq[0][0]
print(f"last step: {len(seq)} kernels, {(max(e[1] for e in seq) - t0) / 1e3:.1f} us from first start to last end")
for s, e, n in seq:
    gap = (s - prev_end) / 1e3
    if gap >= min_gap:
        print(f"  gap {gap:7.1f} us before {short(n)} (at {(s - t0) / 1e3:8.1f} us)")
    busy += e - s
    prev_end = max(prev_end, e)
print(f"busy {busy / 1e3:.1f} us, idle {(prev_end - t0 - busy) / 1e3:.1f} us")
