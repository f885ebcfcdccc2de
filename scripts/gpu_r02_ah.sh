#!/bin/bash
# Session AH: A = HEAD's pruned kernel; B = E2 fetched after E0's last use + the f64 kernel built for 3 waves per SIMD
mkdir -p gpurun_out
export PYTHONPATH=$PWD
LOG=gpurun_out/r02ah.log
: > $LOG
short() { python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(d['dtype'], d['kernels_ms']['spgemm_topn'], round(d['ms_per_step'], 3), d['roofline']['avg_ms'], d['matches'])"; }
for rep in 1 2; do
for dt in f32 f64; do
for v in "SG_HIP_LIB=$PWD/string_grouper_amd/libsg_hip_probeA.so" "SG_X=1" "SG_SYM=0 SG_HIP_LIB=$PWD/string_grouper_amd/libsg_hip_probeA.so" "SG_SYM=0"; do
  echo -n "$dt $v : " | sed "s|$PWD/string_grouper_amd/||" >> $LOG
  env $v timeout 300 python bench.py --dtype $dt --steps 4 --warmup 1 --no-cpu-baseline --no-exact-kernel --no-end-to-end 2>gpurun_out/r02ah_err.log | short >> $LOG 2>&1
done
done
done
cat $LOG
