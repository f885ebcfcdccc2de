#!/bin/bash
# Round 2, session D: whole GPU suite after the tokeniser-domain work + the pad fix, smoke, default bench.
mkdir -p gpurun_out
export PYTHONPATH=$PWD
LOG=gpurun_out/r02d.log
: > $LOG
make -s -C oracle
timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=900 --durations=8 > gpurun_out/r02d_pytest.log 2>&1
echo "pytest exit $?" >> $LOG; tail -60 gpurun_out/r02d_pytest.log >> $LOG
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" >> $LOG 2>&1
echo "== default bench" >> $LOG
timeout 900 python bench.py > gpurun_out/r02d_bench.json 2> gpurun_out/r02d_bench.err
echo "bench exit $?" >> $LOG; cat gpurun_out/r02d_bench.json >> $LOG; tail -5 gpurun_out/r02d_bench.err >> $LOG
cat $LOG
