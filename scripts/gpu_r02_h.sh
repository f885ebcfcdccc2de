#!/bin/bash
# Round 2, session H: full GPU suite (wide rows, K2 properties), family sweep, masked-load / tile experiments.
mkdir -p gpurun_out
export PYTHONPATH=$PWD
LOG=gpurun_out/r02h.log
: > $LOG
make -s -C oracle
echo "== full GPU suite" >> $LOG
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=900 > gpurun_out/r02h_pytest.log 2>&1
echo "exit $?" >> $LOG; tail -30 gpurun_out/r02h_pytest.log >> $LOG
echo "== family sweep" >> $LOG
timeout 900 python scripts/family_sweep.py >> $LOG 2>&1
short() { python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(json.dumps({k: d.get(k) for k in ('value', 'ms_per_step', 'kernels_ms', 'matches')}))"; }
for v in "SG_SYM=1" "SG_SYM=1 SG_PRUNE_MASKED=1" "SG_SYM=0" "SG_SYM=0 SG_PRUNE_MASKED=1" "SG_SYM=1 SG_PRUNE_TILE=11" "SG_SYM=1 SG_PRUNE_TILE=11 SG_PRUNE_MASKED=1"; do
  echo "== bench $v" >> $LOG
  env $v timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-exact-kernel --no-end-to-end 2>/dev/null | short >> $LOG 2>&1
done
cat $LOG
