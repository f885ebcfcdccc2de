#!/bin/bash
mkdir -p gpurun_out
export PYTHONPATH=$PWD
LOG=gpurun_out/run5.log
: > $LOG
make -s -C oracle
echo "== pytest" >> $LOG
timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -p no:cacheprovider --timeout=150 > gpurun_out/pytest5.log 2>&1
tail -15 gpurun_out/pytest5.log >> $LOG
echo "== sweep" >> $LOG
timeout 400 python scripts/kernel_sweep.py 100000,663000 > gpurun_out/sweep5.log 2>&1
grep '"what": "spgemm"' gpurun_out/sweep5.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['n'], d['tile'], d['group'], d['depth'], '%.1f ms' % d['ms_event'], '%.2f TB/s' % d['alg_TBps'])
" >> $LOG
cd /tmp && export TMPDIR=/tmp && cd $OLDPWD
BENCH="python bench.py --steps 2 --warmup 1 --no-cpu-baseline"
echo "== rocprof stats" >> $LOG
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_stats -o k4 -- $BENCH > gpurun_out/prof_stats.out 2>&1
find gpurun_out/prof_stats -name "*kernel_stats*" | head -2 >> $LOG
for f in $(find gpurun_out/prof_stats -name "*kernel_stats.csv" | head -1); do head -12 $f >> $LOG; done
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVES" \
            "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  echo "== pmc pass $i: $ctrs" >> $LOG
  timeout 300 rocprofv3 --pmc $ctrs -d gpurun_out/prof_pmc$i -o k4 -- $BENCH > gpurun_out/prof_pmc$i.out 2>&1
  python scripts/pmc_summary.py gpurun_out/prof_pmc$i >> $LOG 2>&1
done
cat $LOG
