#!/bin/bash
# A/B in one session (same box): the pruned kernel before the wide-row generalisation vs now, both forms, alternating.
mkdir -p gpurun_out
export PYTHONPATH=$PWD
LOG=gpurun_out/r02i.log
: > $LOG
short() { python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(d['kernels_ms']['spgemm_topn'], round(d['ms_per_step'], 3))"; }
for rep in 1 2; do
for lib in string_grouper_amd/libsg_hip_oldk4p.so string_grouper_amd/libsg_hip.so; do
for v in "SG_SYM=1" "SG_SYM=0"; do
  echo -n "$lib $v : " >> $LOG
  env $v SG_HIP_LIB=$PWD/$lib timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-exact-kernel --no-end-to-end 2>/dev/null | short >> $LOG 2>&1
done
done
done
SG_PRUNE_MASKED=1 SG_SYM=1 timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-exact-kernel --no-end-to-end 2>/dev/null | short >> $LOG 2>&1
cat $LOG
