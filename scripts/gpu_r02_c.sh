#!/bin/bash
# Round 2, session C: whole GPU suite (new: reference-made fixtures at 20-30k, 663k all rows, sharded path on a 1-rank
# RCCL group, unseen characters), smoke, the default bench line, the 1-rank distributed bench.
mkdir -p gpurun_out
export PYTHONPATH=$PWD
LOG=gpurun_out/r02c.log
: > $LOG
make -s -C oracle
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider --timeout=900 --durations=8 > gpurun_out/r02c_pytest.log 2>&1
echo "pytest exit $?" >> $LOG; tail -25 gpurun_out/r02c_pytest.log >> $LOG
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" >> $LOG 2>&1
echo "== default bench" >> $LOG
timeout 900 python bench.py > gpurun_out/r02c_bench.json 2> gpurun_out/r02c_bench.err
echo "bench exit $?" >> $LOG; cat gpurun_out/r02c_bench.json >> $LOG; tail -5 gpurun_out/r02c_bench.err >> $LOG
echo "== 1-rank RCCL run of the sharded bench path" >> $LOG
SG_BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-exact-kernel --no-end-to-end > gpurun_out/r02c_bench_dist.json 2> gpurun_out/r02c_bench_dist.err
echo "exit $?" >> $LOG; cat gpurun_out/r02c_bench_dist.json >> $LOG; tail -3 gpurun_out/r02c_bench_dist.err >> $LOG
cat $LOG
