#!/bin/bash
mkdir -p gpurun_out
export PYTHONPATH=$PWD
LOG=gpurun_out/run4.log
: > $LOG
make -s -C oracle
echo "== pytest" >> $LOG
timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -p no:cacheprovider --timeout=150 > gpurun_out/pytest4.log 2>&1
tail -15 gpurun_out/pytest4.log >> $LOG
echo "== sweep" >> $LOG
timeout 400 python scripts/kernel_sweep.py 100000,663000 > gpurun_out/sweep4.log 2>&1
grep spgemm gpurun_out/sweep4.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['n'], d['tile'], d['group'], d['depth'], '%.1f ms' % d['ms_event'], '%.2f TB/s' % d['alg_TBps'])
" >> $LOG
echo "== bench" >> $LOG
timeout 600 python bench.py --steps 3 --warmup 1 > gpurun_out/bench4.json 2> gpurun_out/bench4.err
cat gpurun_out/bench4.json >> $LOG; tail -5 gpurun_out/bench4.err >> $LOG
echo "== smoke" >> $LOG
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" >> $LOG 2>&1
cat $LOG
