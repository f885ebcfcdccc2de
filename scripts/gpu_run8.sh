#!/bin/bash
mkdir -p gpurun_out
export PYTHONPATH=$PWD
LOG=gpurun_out/run8.log
: > $LOG
make -s -C oracle
echo "== pytest (all gpu tests)" >> $LOG
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider --timeout=200 > gpurun_out/pytest8.log 2>&1
tail -25 gpurun_out/pytest8.log >> $LOG
echo "== bench via torchrun, 1 rank, forced distributed path" >> $LOG
SG_BENCH_FORCE_DIST=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 1 --steps 3 --warmup 1 > gpurun_out/bench8_dist.json 2> gpurun_out/bench8_dist.err
cat gpurun_out/bench8_dist.json >> $LOG; tail -8 gpurun_out/bench8_dist.err >> $LOG
echo "== smoke" >> $LOG
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" >> $LOG 2>&1
cat $LOG
