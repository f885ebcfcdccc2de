#!/bin/bash
mkdir -p gpurun_out
export PYTHONPATH=$PWD
LOG=gpurun_out/r02m.log
: > $LOG
timeout 1200 python -m pytest tests -x -q -m gpu -k "not perf" > gpurun_out/r02m_pytest.log 2>&1
echo "pytest exit $?" >> $LOG
tail -3 gpurun_out/r02m_pytest.log >> $LOG
timeout 600 python scripts/sym_sweep.py >> $LOG 2>&1
short() { python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(d['kernels_ms'], round(d['ms_per_step'], 3), d['roofline']['avg_ms'])"; }
for v in "SG_X=0" "SG_X=1"; do
  echo -n "$v : " >> $LOG
  env $v timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-exact-kernel --no-end-to-end 2>/dev/null | short >> $LOG 2>&1
done
cat $LOG
