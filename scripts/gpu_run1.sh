#!/bin/bash
# first GPU session: parity tests + kernel sweep
mkdir -p gpurun_out
export PYTHONPATH=$PWD
make -s -C oracle
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -150 > gpurun_out/pytest2.log
tail -40 gpurun_out/pytest2.log
timeout 600 python scripts/kernel_sweep.py 100000,663000 > gpurun_out/sweep2.log 2>&1
tail -50 gpurun_out/sweep2.log
