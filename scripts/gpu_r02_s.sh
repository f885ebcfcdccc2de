#!/bin/bash
mkdir -p gpurun_out
export PYTHONPATH=$PWD
timeout 900 python -m pytest tests -x -q -m gpu -k "alternative_forms" > gpurun_out/r02s_pytest.log 2>&1
echo "pytest exit $?"; tail -3 gpurun_out/r02s_pytest.log
bash scripts/gpu_r02_l.sh
