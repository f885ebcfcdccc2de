#!/bin/bash
mkdir -p gpurun_out
export PYTHONPATH=$PWD
LOG=gpurun_out/fast.log
: > $LOG
make -s -C oracle
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider --timeout=300 -k "not config4 and not config5" > gpurun_out/fast_pytest.log 2>&1
grep -E "passed|failed|Error" gpurun_out/fast_pytest.log | tail -5 >> $LOG
timeout 300 python scripts/fast_vs_exact.py 100000 >> $LOG 2>&1
timeout 300 python scripts/fast_vs_exact.py 663000 >> $LOG 2>&1
cat $LOG
