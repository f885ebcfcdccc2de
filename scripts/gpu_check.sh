#!/bin/bash
# One GPU session that re-validates the round: all GPU tests, smoke, default bench line.
#   gpurun --timeout 1500 -- 'bash scripts/gpu_check.sh'
mkdir -p gpurun_out
export PYTHONPATH=$PWD
LOG=gpurun_out/check.log
: > $LOG
make -s -C oracle
timeout 1200 python -m pytest tests -m gpu -q -x -p no:cacheprovider --timeout=600 > gpurun_out/check_pytest.log 2>&1
grep -E "passed|failed|Error" gpurun_out/check_pytest.log | tail -5 >> $LOG
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" >> $LOG 2>&1
timeout 900 python bench.py > gpurun_out/check_bench.json 2> gpurun_out/check_bench.err
cat gpurun_out/check_bench.json >> $LOG
cat $LOG
