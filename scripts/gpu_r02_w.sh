#!/bin/bash
# Session W: per-kernel times of the pruned multiply's launches with the heavy launch on / off (rocprofv3 kernel trace)
mkdir -p gpurun_out
export PYTHONPATH=$PWD
LOG=gpurun_out/r02w.log
: > $LOG
cd /tmp && export TMPDIR=/tmp && cd $OLDPWD
BENCH="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-exact-kernel --no-end-to-end"
for v in "SG_PRUNE_HEAVY=off" "SG_PRUNE_HEAVY=0.75" "SG_PRUNE_HEAVY=0.75 SG_SYM=0" "SG_PRUNE_HEAVY=0"; do
  tag=$(echo $v | tr -c 'A-Za-z0-9\n' '_')
  echo "== $v" >> $LOG
  env $v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02w_$tag -o k -- $BENCH > gpurun_out/r02w_$tag.out 2>&1
  for f in $(find gpurun_out/r02w_$tag -name "*kernel_stats.csv" | head -1); do python - $f >> $LOG <<'PY'
import csv, sys, re
for row in csv.DictReader(open(sys.argv[1])):
    n = row["Name"]
    if "pruned" in n or "pairs_" in n:
        m = re.search(r"(\w+)<([^>]*)>", n)
        print(f"{m.group(1)}<{m.group(2)}>  calls {row['Calls']}  avg {float(row['AverageNs'])/1e6:.3f} ms  min {float(row['MinNs'])/1e6:.3f}  max {float(row['MaxNs'])/1e6:.3f}")
PY
  done
  rm -rf gpurun_out/r02w_$tag
done
cat $LOG
