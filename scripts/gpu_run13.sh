#!/bin/bash
mkdir -p gpurun_out
export PYTHONPATH=$PWD
LOG=gpurun_out/run13.log
: > $LOG
make -s -C oracle
timeout 600 python -m pytest tests -m gpu -q -x -p no:cacheprovider --timeout=300 -k "rccl or golden" > gpurun_out/pytest13.log 2>&1
tail -5 gpurun_out/pytest13.log >> $LOG
for mode in strings csr; do
echo "== bench forced dist mode=$mode" >> $LOG
SG_BENCH_DIST_MODE=$mode SG_BENCH_FORCE_DIST=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench13_$mode.json 2> gpurun_out/bench13_$mode.err
python -c "
import json; d=json.load(open('gpurun_out/bench13_$mode.json')); print(d['value'], d['ms_per_step'], d['kernels_ms'], d['roofline']['frac'])" >> $LOG 2>&1
tail -3 gpurun_out/bench13_$mode.err >> $LOG
done
cat $LOG
