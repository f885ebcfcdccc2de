#!/bin/bash
# pruned multiply: native smoke (watchdog build, tight timeouts), then pruned-vs-exact check + timing
mkdir -p gpurun_out
if [ -z "$SKIP_DIAG" ]; then
bash scripts/gpu_diag.sh > gpurun_out/prune_diag.log 2>&1
grep -E "^exit|mismatches|watch|pruned rows|rc=" gpurun_out/prune_diag.log
fi
timeout 400 python scripts/prune_check.py ${1:-20000,200000,663000} ${2:-0.2} ${3:-f32} ${4:-0.003} ${5:-12} ${6:-0.8} > gpurun_out/prune_check.log 2>&1
echo "check exit $?"
grep -E "^\{" gpurun_out/prune_check.log | cut -c1-400
tail -3 gpurun_out/prune_check.log | grep -v "^{" | cut -c1-300
