#!/bin/bash
# Final profile of the round, short form: rocprofv3 kernel stats + PMC passes (separate runs, no trace domains with --pmc)
# of `python bench.py --steps 3 --warmup 1 --no-cpu-baseline [--no-exact-kernel] --no-end-to-end`.
mkdir -p gpurun_out
export PYTHONPATH=$PWD
TAG=${1:-r02b}
LOG=gpurun_out/${TAG}_profile.log
: > $LOG
cd /tmp && export TMPDIR=/tmp && cd $OLDPWD
BENCH="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end"          # incl. one live run of the exact kernel
BENCH_PMC="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-exact-kernel --no-end-to-end"
echo "== rocprofv3 --kernel-trace --stats -- $BENCH" >> $LOG
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_stats -o k -- $BENCH > gpurun_out/${TAG}_stats.out 2>&1
grep '^{' gpurun_out/${TAG}_stats.out | tail -1 >> $LOG
for f in $(find gpurun_out/${TAG}_stats -name "*kernel_stats.csv" | head -1); do cat $f >> $LOG; cp $f gpurun_out/${TAG}_kernel_stats.csv; done
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
            "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  echo "== pmc pass $i: $ctrs" >> $LOG
  timeout 300 rocprofv3 --pmc $ctrs --output-format csv -d gpurun_out/${TAG}_pmc$i -o k -- $BENCH_PMC > gpurun_out/${TAG}_pmc$i.out 2>&1
  python scripts/pmc_summary.py gpurun_out/${TAG}_pmc$i | head -12 >> $LOG 2>&1
done
rm -rf gpurun_out/${TAG}_stats gpurun_out/${TAG}_pmc[0-9]
tail -60 $LOG | cut -c1-250
