#!/bin/bash
mkdir -p gpurun_out
export PYTHONPATH=$PWD
timeout 600 python scripts/l2_experiment.py > gpurun_out/l2.log 2>&1
cat gpurun_out/l2.log
