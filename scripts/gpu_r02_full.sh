#!/bin/bash
# Round 2: whole GPU suite, smoke, default bench (the round-end validation; output tag r02i).
mkdir -p gpurun_out
export PYTHONPATH=$PWD
LOG=gpurun_out/r02i.log
: > $LOG
make -s -C oracle
timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=900 --durations=8 > gpurun_out/r02i_pytest.log 2>&1
echo "pytest exit $?" >> $LOG; tail -60 gpurun_out/r02i_pytest.log >> $LOG
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" >> $LOG 2>&1
echo "== default bench" >> $LOG
timeout 900 python bench.py > gpurun_out/r02i_bench.json 2> gpurun_out/r02i_bench.err
echo "bench exit $?" >> $LOG; cat gpurun_out/r02i_bench.json >> $LOG; tail -5 gpurun_out/r02i_bench.err >> $LOG
cat $LOG
