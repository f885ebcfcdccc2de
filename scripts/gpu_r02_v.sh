#!/bin/bash
# Round 2, session V: (1) parity of the two changes -- rows for the exact kernel inside the self-join form, the heavy
# launch of the pruned kernel; (2) A/B of the heavy launch at 663 k (SG_PRUNE_HEAVY=off is the previous behaviour).
mkdir -p gpurun_out
export PYTHONPATH=$PWD
LOG=gpurun_out/r02v.log
: > $LOG
timeout 900 python -m pytest tests -x -q -m gpu -p no:cacheprovider -k "heavy or beyond_64 or row_ranges or every_tuning or hubs or pair_list or equals_exact_and_oracle" > gpurun_out/r02v_pytest.log 2>&1
echo "pytest exit $?" >> $LOG; tail -15 gpurun_out/r02v_pytest.log | cut -c1-300 >> $LOG
short() { python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(d['kernels_ms']['spgemm_topn'], round(d['ms_per_step'], 3), d['roofline']['avg_ms'], d['pruning'], d['matches'])"; }
for v in "SG_PRUNE_HEAVY=off" "SG_PRUNE_HEAVY=0.75" "SG_PRUNE_HEAVY=0.6" "SG_PRUNE_HEAVY=0.9" "SG_PRUNE_HEAVY=0.5" "SG_PRUNE_HEAVY=off SG_SYM=0" "SG_PRUNE_HEAVY=0.75 SG_SYM=0"; do
  echo -n "$v : " >> $LOG
  env $v timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-exact-kernel --no-end-to-end 2>gpurun_out/r02v_err.log | short >> $LOG 2>&1
done
cat $LOG
