#!/bin/bash
mkdir -p gpurun_out
export PYTHONPATH=$PWD
LOG=gpurun_out/run6.log
: > $LOG
make -s -C oracle
echo "== pytest" >> $LOG
timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -p no:cacheprovider --timeout=150 > gpurun_out/pytest6.log 2>&1
tail -15 gpurun_out/pytest6.log >> $LOG
echo "== sweep" >> $LOG
timeout 400 python scripts/kernel_sweep.py 100000,663000 > gpurun_out/sweep6.log 2>&1
grep '"what": "spgemm"' gpurun_out/sweep6.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['n'], d['tile'], d['group'], d['depth'], '%.1f ms' % d['ms_event'], '%.2f TB/s' % d['alg_TBps'])
" >> $LOG
cd /tmp && export TMPDIR=/tmp && cd $OLDPWD
BENCH="python bench.py --steps 2 --warmup 1 --no-cpu-baseline"
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVES"; do
  i=$((i+1))
  echo "== pmc pass $i: $ctrs" >> $LOG
  timeout 300 rocprofv3 --pmc $ctrs --output-format csv -d gpurun_out/prof6_pmc$i -o k4 -- $BENCH > gpurun_out/prof6_pmc$i.out 2>&1
  python scripts/pmc_summary.py gpurun_out/prof6_pmc$i >> $LOG 2>&1
done
cat $LOG
