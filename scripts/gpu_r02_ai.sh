#!/bin/bash
# Session AI (f64): A = HEAD; C = E2 fetched after E0's last use; D = C without the position -> row mapping (timing probe);
# B = C built for three waves per SIMD
mkdir -p gpurun_out
export PYTHONPATH=$PWD
LOG=gpurun_out/r02ai.log
: > $LOG
short() { python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(d['dtype'], d['kernels_ms']['spgemm_topn'], round(d['ms_per_step'], 3), d['roofline']['avg_ms'], d['matches'])"; }
for rep in 1 2; do
for lib in libsg_hip_probeA.so libsg_hip.so libsg_hip_probeD.so libsg_hip_probeB.so; do
for v in "SG_SYM=1" "SG_SYM=0"; do
  echo -n "f64 $lib $v : " >> $LOG
  env $v SG_HIP_LIB=$PWD/string_grouper_amd/$lib timeout 300 python bench.py --dtype f64 --steps 4 --warmup 1 --no-cpu-baseline --no-exact-kernel --no-end-to-end 2>gpurun_out/r02ai_err.log | short >> $LOG 2>&1
done
done
done
cat $LOG
