"""The launches of the LAST step of the hot path in order, from a rocprofv3 kernel trace: start, duration, gap to the launch
before.   python scripts/ktrace_step.py <kernel_trace.csv>"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda e: e[0])


def short(n):
    m = re.match(r"(?:void )?([\w:]+)", n)
    return (m.group(1) if m else n)[:44] + ("<" + n.split("<", 1)[1][:40] if "<" in n else "")


starts = [i for i, e in enumerate(ev) if "tokenize_short_kernel" in e[2]]
seq = ev[starts[-2]:starts[-1]] if len(starts) > 1 else ev[starts[-1]:]
t0, prev = seq[0][0], seq[0][0]
for k, (s, e, n) in enumerate(seq):
    print(f"{k:3d} at {(s - t0) / 1e3:8.1f} us  {(e - s) / 1e3:8.1f} us  gap {(s - prev) / 1e3:6.1f}  {short(n)}")
    prev = max(prev, e)
print(f"{len(seq)} launches, {(prev - t0) / 1e3:.1f} us, busy {sum(e - s for s, e, _ in seq) / 1e3:.1f} us")
