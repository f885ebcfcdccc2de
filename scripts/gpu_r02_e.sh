#!/bin/bash
# Round 2, session E: the 663k test again, K4p vs K4 on other data families, and the round's profile:
# rocprofv3 --kernel-trace --stats + separate --pmc passes of the default bench workload (symmetric self-join form).
mkdir -p gpurun_out
export PYTHONPATH=$PWD
LOG=gpurun_out/r02e.log
: > $LOG
make -s -C oracle
timeout 600 python -m pytest tests/test_parity_gpu.py -q -p no:cacheprovider --timeout=600 -k "headline_663k or any_length or whole_domain or wide_keys" > gpurun_out/r02e_pytest.log 2>&1
echo "pytest exit $?" >> $LOG; tail -4 gpurun_out/r02e_pytest.log >> $LOG
echo "== family sweep" >> $LOG
timeout 900 python scripts/family_sweep.py >> $LOG 2>&1
cd /tmp && export TMPDIR=/tmp && cd $OLDPWD
BENCH="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end"          # incl. one live run of the exact kernel
BENCH_PMC="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-exact-kernel --no-end-to-end"
echo "== rocprofv3 --kernel-trace --stats" >> $LOG
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02e_stats -o k -- $BENCH > gpurun_out/r02e_stats.out 2>&1
for f in $(find gpurun_out/r02e_stats -name "*kernel_stats.csv" | head -1); do cat $f >> $LOG; cp $f gpurun_out/r02_kernel_stats.csv; done
rm -rf gpurun_out/r02e_stats
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVES" \
            "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
            "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"; do
  i=$((i+1))
  echo "== pmc pass $i: $ctrs" >> $LOG
  timeout 240 rocprofv3 --pmc $ctrs --output-format csv -d gpurun_out/r02e_pmc$i -o k -- $BENCH_PMC > gpurun_out/r02e_pmc$i.out 2>&1
  python scripts/pmc_summary.py gpurun_out/r02e_pmc$i 2>&1 | head -22 >> $LOG
  rm -rf gpurun_out/r02e_pmc$i
done
cat $LOG
