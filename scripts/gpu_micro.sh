#!/bin/bash
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tests/native/lds_microbench.cpp -o /tmp/lds_microbench && timeout 120 /tmp/lds_microbench > gpurun_out/lds_microbench.log 2>&1
cat gpurun_out/lds_microbench.log
