#!/bin/bash
# Session Z: where do 3.6 ms come from after the larger pair list + the early-abort check? (same box A/B)
mkdir -p gpurun_out
export PYTHONPATH=$PWD
LOG=gpurun_out/r02z.log
: > $LOG
short() { python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(d['kernels_ms']['spgemm_topn'], round(d['ms_per_step'], 3), d['roofline']['avg_ms'], d['matches'])"; }
for rep in 1 2; do
for v in "SG_X=1" "SG_SYM_PAIR_CAP=6352000" "SG_HIP_LIB=$PWD/string_grouper_amd/libsg_hip_probeNA.so" "SG_HIP_LIB=$PWD/string_grouper_amd/libsg_hip_probeNA.so SG_SYM_PAIR_CAP=6352000"; do
  echo -n "$v : " >> $LOG
  env $v timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-exact-kernel --no-end-to-end 2>gpurun_out/r02z_err.log | short >> $LOG 2>&1
done
done
cat $LOG
