#!/bin/bash
# K4p with exec-masked batch loads (inline assembly, hand-written waits): parity subset on that build, then A/B.
mkdir -p gpurun_out
export PYTHONPATH=$PWD
LOG=gpurun_out/r02t.log
: > $LOG
SG_HIP_LIB=$PWD/string_grouper_amd/libsg_hip_masked.so timeout 1200 python -m pytest tests -x -q -m gpu -k "prun or hub or wide or headline or selfjoin or 100k or spgemm or pilot or blocked" > gpurun_out/r02t_pytest.log 2>&1
echo "pytest (masked build) exit $?" >> $LOG
tail -3 gpurun_out/r02t_pytest.log >> $LOG
short() { python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(d['kernels_ms']['spgemm_topn'], round(d['ms_per_step'], 3), d['roofline']['avg_ms'])"; }
for rep in 1 2; do
for lib in libsg_hip.so libsg_hip_masked.so; do
for v in "SG_SYM=1" "SG_SYM=0"; do
  echo -n "$lib $v : " >> $LOG
  env $v SG_HIP_LIB=$PWD/string_grouper_amd/$lib timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-exact-kernel --no-end-to-end 2>/dev/null | short >> $LOG 2>&1
done
done
done
cat $LOG
