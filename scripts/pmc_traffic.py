"""Memory-side traffic of the multiply kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: separate runs, no
trace domains), per launch, corrected as /opt/skills/guides/MI355X_MICROARCH.md (HBM) prescribes for gfx950: FETCH_SIZE
tallies 128-byte memory-side requests at 64 B, so read bytes = 2 x FETCH_SIZE; WRITE_SIZE is taken as it is
(uncalibrated, 0.5 % of the total here).  Infinity-Cache hits are counted, not excluded.

    python scripts/pmc_traffic.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <out.json> [rows] [dtype] [kernel tag]

Writes the JSON that bench.py reads for roofline.traffic (profiles/k4_traffic.json is a copy of such a file)."""
import csv
import glob
import json
import os
import sys


def per_launch(d, counter):
    vals = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") == counter and "spgemm_topn_pruned_kernel" in row.get("Kernel_Name", "") \
                        and "Lb1E" not in row.get("Kernel_Name", "xx"):
                    vals.append(float(row["Counter_Value"]))
    # the two launches of one multiply (rows of up to 64 non-zeros, then the wide ones): the large values are the first
    big = [v for v in vals if v > 0.05 * max(vals)] if vals else []
    return (sum(big) / len(big), len(big)) if big else (0.0, 0)


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    rows = int(sys.argv[4]) if len(sys.argv) > 4 else 663000
    dtype = sys.argv[5] if len(sys.argv) > 5 else "f32"
    tag = sys.argv[6] if len(sys.argv) > 6 else "K4p-sym"
    fetch_kb, n1 = per_launch(fetch_dir, "FETCH_SIZE")
    write_kb, n2 = per_launch(write_dir, "WRITE_SIZE")
    read_bytes = 2.0 * fetch_kb * 1024.0
    write_bytes = write_kb * 1024.0
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from string_grouper_amd._provenance import kernel_source_sha
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    res = {"workload_rows": rows, "dtype": dtype, "kernel": tag, "source_sha": kernel_source_sha(root),
           "traffic_bytes_per_launch_raw": read_bytes + write_bytes,
           "read_bytes": read_bytes, "write_bytes": write_bytes, "launches_averaged": [n1, n2],
           "source": "scripts/pmc_traffic.py over two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, no trace "
                     "domains) of `python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-exact-kernel --no-end-to-end --no-side-runs`",
           "note": "read = 2 x FETCH_SIZE (gfx950 tallies 128-byte memory-side requests at 64 B: MI355X_MICROARCH.md, HBM); "
                   "WRITE_SIZE uncalibrated; Infinity-Cache hits included"}
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    print(f"FETCH_SIZE {fetch_kb:.4g} KB x2 = {read_bytes / 1e9:.2f} GB read, WRITE_SIZE {write_bytes / 1e9:.2f} GB "
          f"=> {(read_bytes + write_bytes) / 1e9:.2f} GB per launch ({n1}/{n2} launches)")


if __name__ == "__main__":
    main()
