#!/bin/bash
# Round 2, session A: K4p v9 (static 4-deep batch pipeline, integer filter, symmetric self-join mode).
#   1. watchdog build of the library -> the pruned-kernel parity tests (a runaway loop winds down instead of hanging)
#   2. the whole GPU suite on the shipped library
#   3. bench: symmetric mode / one-sided, waves per CU
mkdir -p gpurun_out /tmp/wd
export PYTHONPATH=$PWD
LOG=gpurun_out/r02a.log
: > $LOG
make -s -C oracle
HIPCC=/opt/rocm/bin/hipcc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function"
( cd string_grouper_amd/csrc
  for f in sg_api sg_postings sg_vectorize sg_matchlist sg_reduce; do cp $f.o /tmp/wd/$f.o; done
  $HIPCC $FLAGS -DSG_WATCHDOG -c sg_spgemm_topn.hip -o /tmp/wd/sg_spgemm_topn.o &
  $HIPCC $FLAGS -DSG_WATCHDOG -c sg_spgemm_pruned.hip -o /tmp/wd/sg_spgemm_pruned.o &
  wait
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o /tmp/wd/libsg_hip.so /tmp/wd/*.o ) >> $LOG 2>&1
echo "== watchdog build: pruned-kernel tests" >> $LOG
SG_HIP_LIB=/tmp/wd/libsg_hip.so timeout 600 python -m pytest tests/test_parity_gpu.py -q -x -p no:cacheprovider --timeout=300 \
   -k "pruned or selfjoin or hubs or ties or edge or twice or 100k" > gpurun_out/r02a_wd.log 2>&1
echo "exit $?" >> $LOG; tail -15 gpurun_out/r02a_wd.log >> $LOG
if ! grep -q " passed" gpurun_out/r02a_wd.log || grep -q "failed\|Timeout" gpurun_out/r02a_wd.log; then
  echo "WATCHDOG RUN NOT CLEAN: skipping the rest" >> $LOG; cat $LOG; exit 1
fi
echo "== full GPU suite, SG_SYM default" >> $LOG
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider --timeout=600 > gpurun_out/r02a_pytest.log 2>&1
echo "exit $?" >> $LOG; tail -8 gpurun_out/r02a_pytest.log >> $LOG
echo "== pruned tests with SG_SYM=0" >> $LOG
SG_SYM=0 timeout 600 python -m pytest tests/test_parity_gpu.py -q -x -p no:cacheprovider --timeout=300 -k "pruned or selfjoin or hubs or 100k" > gpurun_out/r02a_nosym.log 2>&1
echo "exit $?" >> $LOG; tail -4 gpurun_out/r02a_nosym.log >> $LOG
echo "== bench default (symmetric), with cpu baseline + exact kernel" >> $LOG
timeout 900 python bench.py --steps 5 --warmup 1 > gpurun_out/r02a_bench_sym.json 2> gpurun_out/r02a_bench_sym.err
cat gpurun_out/r02a_bench_sym.json >> $LOG; tail -3 gpurun_out/r02a_bench_sym.err >> $LOG
for v in "SG_SYM=0" "SG_SYM=0 SG_PRUNE_WAVES_PER_CU=12" "SG_SYM=1 SG_PRUNE_WAVES_PER_CU=12" "SG_SYM=0 SG_PRUNE_DELTA=0.08" "SG_SYM=1 SG_PRUNE_DELTA=0.08" "SG_SYM=0 SG_PRUNE_TILE=13" "SG_SYM=1 SG_PRUNE_TILE=13"; do
  echo "== bench $v" >> $LOG
  env $v timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-exact-kernel 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(json.dumps({k: d[k] for k in ('value', 'ms_per_step', 'kernels_ms', 'matches', 'pruning')}))" >> $LOG 2>&1
done
echo "== bench f64 (symmetric)" >> $LOG
timeout 300 python bench.py --steps 5 --warmup 1 --dtype f64 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(json.dumps({k: d.get(k) for k in ('value', 'ms_per_step', 'kernels_ms', 'matches', 'pruning', 'exact_kernel')}))" >> $LOG 2>&1
cat $LOG
