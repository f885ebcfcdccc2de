#!/bin/bash
# Round 2, session G: the wide-row launch of K4p (rows with 65..128 non-zeros), the pilot, the family guards; family sweep; bench.
mkdir -p gpurun_out /tmp/wd
export PYTHONPATH=$PWD
LOG=gpurun_out/r02g.log
: > $LOG
make -s -C oracle
HIPCC=/opt/rocm/bin/hipcc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function"
( cd string_grouper_amd/csrc
  for f in sg_api sg_postings sg_vectorize sg_matchlist sg_reduce sg_sortvocab; do cp $f.o /tmp/wd/$f.o; done
  $HIPCC $FLAGS -DSG_WATCHDOG -c sg_spgemm_topn.hip -o /tmp/wd/sg_spgemm_topn.o &
  $HIPCC $FLAGS -DSG_WATCHDOG -c sg_spgemm_pruned.hip -o /tmp/wd/sg_spgemm_pruned.o &
  wait
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o /tmp/wd/libsg_hip.so /tmp/wd/*.o ) >> $LOG 2>&1
echo "== watchdog build: pruned-kernel tests" >> $LOG
SG_HIP_LIB=/tmp/wd/libsg_hip.so timeout 600 python -m pytest tests/test_parity_gpu.py -q -x -p no:cacheprovider --timeout=300 \
   -k "pruned or selfjoin or hubs or ties or edge or twice or 100k or beyond_64 or pilot" > gpurun_out/r02g_wd.log 2>&1
echo "exit $?" >> $LOG; tail -15 gpurun_out/r02g_wd.log >> $LOG
if ! grep -q " passed" gpurun_out/r02g_wd.log || grep -q "failed\|Timeout" gpurun_out/r02g_wd.log; then
  echo "WATCHDOG RUN NOT CLEAN: skipping the rest" >> $LOG; cat $LOG; exit 1
fi
echo "== full GPU suite" >> $LOG
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=900 > gpurun_out/r02g_pytest.log 2>&1
echo "exit $?" >> $LOG; tail -30 gpurun_out/r02g_pytest.log >> $LOG
echo "== family sweep" >> $LOG
timeout 900 python scripts/family_sweep.py >> $LOG 2>&1
echo "== bench" >> $LOG
timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/r02g_bench.json 2> gpurun_out/r02g_bench.err
python -c "
import json
d = json.load(open('gpurun_out/r02g_bench.json'))
print(json.dumps({k: d.get(k) for k in ('value', 'ms_per_step', 'kernels_ms', 'matches', 'pruning', 'end_to_end')}))" >> $LOG 2>&1
cat $LOG
