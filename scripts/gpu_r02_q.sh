#!/bin/bash
# component probes of K4p (wrong results, timing only): A no survivor handling, B + no LDS, C + no loads, D A + no loads
mkdir -p gpurun_out
export PYTHONPATH=$PWD
LOG=gpurun_out/r02q.log
: > $LOG
short() { python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(d['kernels_ms']['spgemm_topn'], round(d['ms_per_step'], 3), d['roofline']['avg_ms'])"; }
for lib in libsg_hip.so libsg_hip_probeA.so libsg_hip_probeB.so libsg_hip_probeC.so libsg_hip_probeD.so; do
for w in 16 8; do
  echo -n "$lib waves/CU $w SYM=1 : " >> $LOG
  env SG_SYM=1 SG_PRUNE_WAVES_PER_CU=$w SG_HIP_LIB=$PWD/string_grouper_amd/$lib timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-exact-kernel --no-end-to-end 2>/dev/null | short >> $LOG 2>&1
done
done
cat $LOG
