#!/bin/bash
mkdir -p gpurun_out
export PYTHONPATH=$PWD
LOG=gpurun_out/run12.log
: > $LOG
make -s -C oracle
echo "== pytest" >> $LOG
timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -k "not config4 and not config5" -p no:cacheprovider --timeout=150 > gpurun_out/pytest12.log 2>&1
tail -15 gpurun_out/pytest12.log >> $LOG
echo "== sweep" >> $LOG
timeout 400 python scripts/kernel_sweep.py 100000,663000 > gpurun_out/sweep12.log 2>&1
grep '"what": "spgemm"' gpurun_out/sweep12.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['n'], d['tile'], d['group'], d['depth'], '%.1f ms' % d['ms_event'], '%.2f TB/s' % d['alg_TBps'])
" >> $LOG
cat $LOG
