#!/bin/bash
mkdir -p gpurun_out
export PYTHONPATH=$PWD
timeout 1200 python -m pytest tests -x -q -m gpu -k "row_ranges or rccl or plumbing or distributed or pair_list" > gpurun_out/r02u_pytest.log 2>&1
echo "pytest exit $?"; tail -25 gpurun_out/r02u_pytest.log | cut -c1-220
