#!/bin/bash
mkdir -p gpurun_out
export PYTHONPATH=$PWD
LOG=gpurun_out/run10.log
: > $LOG
make -s -C oracle
timeout 600 python -m pytest tests -m gpu -q -x -p no:cacheprovider --timeout=150 > gpurun_out/pytest10.log 2>&1
tail -5 gpurun_out/pytest10.log >> $LOG
cd /tmp && export TMPDIR=/tmp && cd $OLDPWD
BENCH="python bench.py --steps 2 --warmup 1 --no-cpu-baseline"
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVES" \
            "SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INSTS_VSKIPPED SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  echo "== pmc pass $i: $ctrs" >> $LOG
  timeout 300 rocprofv3 --pmc $ctrs --output-format csv -d gpurun_out/p10_pmc$i -o k -- $BENCH > gpurun_out/p10_pmc$i.out 2>&1
  python scripts/pmc_summary.py gpurun_out/p10_pmc$i | head -10 >> $LOG 2>&1
  tail -3 gpurun_out/p10_pmc$i.out >> $LOG
done
rm -rf gpurun_out/p10_pmc[0-9]
cat $LOG
