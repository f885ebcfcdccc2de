#!/bin/bash
# occupancy probe of K4p: fewer waves per CU -> proportional slowdown means latency bound
mkdir -p gpurun_out
export PYTHONPATH=$PWD
LOG=gpurun_out/r02o.log
: > $LOG
short() { python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(d['kernels_ms']['spgemm_topn'], round(d['ms_per_step'], 3), d['roofline']['avg_ms'])"; }
for w in 16 12 8 4; do
for v in "SG_SYM=1" "SG_SYM=0"; do
  echo -n "waves/CU $w $v : " >> $LOG
  env $v SG_PRUNE_WAVES_PER_CU=$w timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-exact-kernel --no-end-to-end 2>/dev/null | short >> $LOG 2>&1
done
done
cat $LOG
