#!/bin/bash
# Session AL (f64 self-join kernel over this session's commits, SG_PERMUTE=0 for all): where did 14.7 -> 16.2 ms come from?
mkdir -p gpurun_out
export PYTHONPATH=$PWD
LOG=gpurun_out/r02al.log
: > $LOG
short() { python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(d['dtype'], d['kernels_ms']['spgemm_topn'], round(d['ms_per_step'], 3), d['roofline']['avg_ms'], d['matches'])"; }
for rep in 1 2; do
for lib in libsg_hip_probe_958af57.so libsg_hip_probe_7610a65.so libsg_hip.so; do
for dt in f64 f32; do
  echo -n "$dt $lib : " >> $LOG
  env SG_PERMUTE=0 SG_HIP_LIB=$PWD/string_grouper_amd/$lib timeout 300 python bench.py --dtype $dt --steps 4 --warmup 1 --no-cpu-baseline --no-exact-kernel --no-end-to-end 2>gpurun_out/r02al_err.log | short >> $LOG 2>&1
done
done
done
cat $LOG
