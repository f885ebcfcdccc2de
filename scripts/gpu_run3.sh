#!/bin/bash
# GPU session 3: reproduce the v0 hang under a tight timeout, validate the restructured kernel, sweep.
mkdir -p gpurun_out
export PYTHONPATH=$PWD
LOG=gpurun_out/run3.log
: > $LOG
/opt/rocm/bin/hipcc -O2 tests/native/k4_smoke.cpp -Iinclude -Lstring_grouper_amd -lsg_hip -ldl -o /tmp/k4_smoke >> $LOG 2>&1
echo "== v0 (old build) native smoke" >> $LOG
LD_LIBRARY_PATH=$PWD/build/v0 timeout 25 /tmp/k4_smoke 2000 3000 >> $LOG 2>&1; echo "exit $?" >> $LOG
echo "== v1 native smoke" >> $LOG
for args in "2000 3000" "20000 30000"; do
  LD_LIBRARY_PATH=$PWD/string_grouper_amd timeout 40 /tmp/k4_smoke $args >> $LOG 2>&1; echo "exit $?" >> $LOG
done
make -s -C oracle
echo "== pytest" >> $LOG
timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -p no:cacheprovider --timeout=150 > gpurun_out/pytest3.log 2>&1
tail -15 gpurun_out/pytest3.log >> $LOG
echo "== sweep" >> $LOG
timeout 500 python scripts/kernel_sweep.py 100000,663000 > gpurun_out/sweep3.log 2>&1
tail -45 gpurun_out/sweep3.log >> $LOG
cat $LOG
