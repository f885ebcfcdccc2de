#!/bin/bash
# K1 (rank-sort short-string tokeniser) + K2 (sixteen lanes per row): parity suite, then step times.
mkdir -p gpurun_out
export PYTHONPATH=$PWD
LOG=gpurun_out/r02j.log
: > $LOG
timeout 1200 python -m pytest tests -x -q -m gpu -k "not perf" > gpurun_out/r02j_pytest.log 2>&1
echo "pytest exit $?" >> $LOG
tail -5 gpurun_out/r02j_pytest.log >> $LOG
short() { python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(d['kernels_ms'], round(d['ms_per_step'], 3), d['parity_on_sample'] if 'parity_on_sample' in d else '')"; }
for v in "SG_X=0" "SG_PRUNE_BIG_TICKETS=0" "SG_SYM=0" "SG_SYM=0 SG_PRUNE_BIG_TICKETS=0"; do
  echo -n "$v : " >> $LOG
  env $v timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-exact-kernel --no-end-to-end 2>/dev/null | short >> $LOG 2>&1
done
cat $LOG
