#!/bin/bash
# Round 2, session B: K4p v9b (one LDS wait per tile) -- parity subset, bench, PMC passes incl. the texture path.
mkdir -p gpurun_out
export PYTHONPATH=$PWD
LOG=gpurun_out/r02b.log
: > $LOG
make -s -C oracle
echo "== pruned-kernel tests (SYM default, then SG_SYM=0)" >> $LOG
timeout 600 python -m pytest tests/test_parity_gpu.py -q -x -p no:cacheprovider --timeout=300 -k "pruned or selfjoin or hubs or ties or edge or twice or 100k" > gpurun_out/r02b_t1.log 2>&1
echo "exit $?" >> $LOG; tail -3 gpurun_out/r02b_t1.log >> $LOG
SG_SYM=0 timeout 600 python -m pytest tests/test_parity_gpu.py -q -x -p no:cacheprovider --timeout=300 -k "pruned or selfjoin or hubs or 100k" > gpurun_out/r02b_t2.log 2>&1
echo "exit $?" >> $LOG; tail -3 gpurun_out/r02b_t2.log >> $LOG
if grep -q "failed\|Timeout\|rror" gpurun_out/r02b_t1.log gpurun_out/r02b_t2.log; then echo "TESTS NOT CLEAN" >> $LOG; cat $LOG; exit 1; fi
short() { python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(json.dumps({k: d.get(k) for k in ('value', 'ms_per_step', 'kernels_ms', 'matches', 'pruning')}))"; }
for v in "SG_SYM=1" "SG_SYM=0" "SG_SYM=1 SG_PRUNE_DELTA=0.08" "SG_SYM=0 SG_PRUNE_FREQ=0.002" "SG_SYM=0 SG_PRUNE_FREQ=0.008"; do
  echo "== bench $v" >> $LOG
  env $v timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-exact-kernel 2>/dev/null | short >> $LOG 2>&1
done
cd /tmp && export TMPDIR=/tmp && cd $OLDPWD
echo "== counters available (TA / TCP / TD / SQ vmem)" >> $LOG
rocprofv3 -L 2>/dev/null | grep -oE "\b(TA|TCP|TD|TCC|SQ|SQC|GRBM)_[A-Za-z0-9_]+" | sort -u | tr '\n' ' ' | fold -w 200 >> $LOG
echo >> $LOG
BENCH_PMC="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-exact-kernel"
i=0
for mode in 0 1; do
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVES" \
            "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM_RD SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" \
            "TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
            "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
            "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
            "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum"; do
  i=$((i+1))
  if [ $mode = 1 ] && [ $i -gt 10 ]; then continue; fi      # symmetric mode: the two SQ passes only
  echo "== pmc pass $i (SG_SYM=$mode): $ctrs" >> $LOG
  SG_SYM=$mode timeout 240 rocprofv3 --pmc $ctrs --output-format csv -d gpurun_out/r02b_pmc$i -o k -- $BENCH_PMC > gpurun_out/r02b_pmc$i.out 2>&1
  python scripts/pmc_summary.py gpurun_out/r02b_pmc$i 2>&1 | grep -A9 "spgemm_topn_pruned" | head -10 >> $LOG
  rm -rf gpurun_out/r02b_pmc$i
done
done
echo "== rocprofv3 --kernel-trace --stats (SYM default)" >> $LOG
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02b_stats -o k -- $BENCH_PMC > gpurun_out/r02b_stats.out 2>&1
for f in $(find gpurun_out/r02b_stats -name "*kernel_stats.csv" | head -1); do head -12 $f >> $LOG; cp $f gpurun_out/r02b_kernel_stats.csv; done
rm -rf gpurun_out/r02b_stats
cat $LOG
