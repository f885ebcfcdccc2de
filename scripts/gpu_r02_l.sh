#!/bin/bash
# per-kernel times of one bench run (rocprofv3 kernel trace)
mkdir -p gpurun_out
export PYTHONPATH=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_l
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_l -o r02l -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-exact-kernel --no-end-to-end > $GRAFT_REPO_ROOT/gpurun_out/r02l_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/r02l.err
f=$(find /tmp/prof_l -name "*kernel_stats.csv" | head -1)
cp "$f" $GRAFT_REPO_ROOT/gpurun_out/r02l_kernel_stats.csv
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:40]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>4s} avg_us {float(r['AverageNs'])/1e3:10.1f} tot_ms {float(r['TotalDurationNs'])/1e6:9.2f}")
PY
