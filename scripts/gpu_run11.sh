#!/bin/bash
mkdir -p gpurun_out
export PYTHONPATH=$PWD
make -s -C oracle
timeout 1200 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -s -p no:cacheprovider -k "config4 or config5" > gpurun_out/pytest11.log 2>&1
tail -30 gpurun_out/pytest11.log
