#!/bin/bash
# One GPU session = one gpurun call.  usage: scripts/gpu_session.sh <name> <step> [<step> ...]
#   steps: tests            pytest -m gpu (whole GPU suite)        -> gpurun_out/<name>_pytest.log
#          tests:<expr>     pytest -m gpu -k <expr>
#          smoke            __graft_entry__.smoke()
#          bench[:args]     python bench.py <args> (no CPU baseline unless asked) -> gpurun_out/<name>_bench*.json
#          ab:<ENV=a,b>[:bench args]   the bench line once per value of ENV (kernel time, step time, matches)
#          prof[:args]      rocprofv3 --kernel-trace --stats of the bench command -> gpurun_out/<name>_kernel_stats.csv
#          pmc[:args]       the HBM traffic passes (FETCH_SIZE / WRITE_SIZE, separate runs, no trace domains)
#          py:<script args> python <script args> -> gpurun_out/<name>_<script>.log
#          sh:<command>     any shell command -> gpurun_out/<name>_sh<n>.log
# Everything a step prints goes to gpurun_out/<name>.log; what is worth keeping is copied into profiles/ by hand.
name=$1; shift
mkdir -p gpurun_out
export PYTHONPATH=$PWD
LOG=gpurun_out/${name}.log
: > $LOG
short() { python -c "
import sys, json
for line in sys.stdin:
    if not line.startswith('{'): continue
    d = json.loads(line)
    r = d.get('roofline', {})
    print(d['dtype'], 'kernel_ms', r.get('avg_ms'), 'frac', round(r.get('frac', 0), 4), 'step_ms', round(d['ms_per_step'], 3), 'kernels', d.get('kernels_ms'), 'matches', d.get('matches'), 'pruning', d.get('pruning'), 'identical', (d.get('exact_kernel') or {}).get('pruned_result_identical'))"; }
n=0
for step in "$@"; do
  n=$((n+1))
  kind=${step%%:*}; arg=""; [[ "$step" == *:* ]] && arg=${step#*:}
  echo "=== [$n] $step" >> $LOG
  case $kind in
    tests)
      if [ -n "$arg" ]; then timeout 1500 python -m pytest tests -m gpu -x -q -k "$arg" > gpurun_out/${name}_pytest$n.log 2>&1
      else timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/${name}_pytest$n.log 2>&1; fi
      echo "rc=$?" >> $LOG; tail -5 gpurun_out/${name}_pytest$n.log >> $LOG ;;
    smoke)
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" >> $LOG 2>&1; echo "rc=$?" >> $LOG ;;
    bench)
      timeout 900 python bench.py $arg > gpurun_out/${name}_bench$n.json 2> gpurun_out/${name}_bench$n.err
      echo "rc=$?" >> $LOG; short < gpurun_out/${name}_bench$n.json >> $LOG 2>&1 ;;
    ab)
      spec=${arg%%:*}; bargs="--steps 6 --warmup 2 --no-cpu-baseline --no-exact-kernel --no-end-to-end --no-side-runs"; [[ "$arg" == *:* ]] && bargs=${arg#*:}
      var=${spec%%=*}; vals=${spec#*=}
      for rep in 1 2; do for v in ${vals//,/ }; do
        echo -n "$var=$v : " >> $LOG
        env $var=$v timeout 600 python bench.py $bargs 2> gpurun_out/${name}_ab$n.err | short >> $LOG 2>&1
      done; done ;;
    prof)
      ( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/${name}_prof -o run -- python $OLDPWD/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-exact-kernel --no-end-to-end --no-side-runs $arg > $OLDPWD/gpurun_out/${name}_prof.json 2> $OLDPWD/gpurun_out/${name}_prof.err )
      echo "rc=$?" >> $LOG
      f=$(find gpurun_out/${name}_prof -name "*kernel_stats.csv" | head -1)
      [ -n "$f" ] && cp $f gpurun_out/${name}_kernel_stats$n.csv && head -12 $f | cut -c1-200 >> $LOG
      rm -rf gpurun_out/${name}_prof
      short < gpurun_out/${name}_prof.json >> $LOG 2>&1 ;;
    pmc)
      for c in FETCH_SIZE WRITE_SIZE; do
        ( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --pmc $c --output-format csv -d $OLDPWD/gpurun_out/${name}_pmc${n}_$c -o run -- python $OLDPWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-exact-kernel --no-end-to-end --no-side-runs $arg > /dev/null 2> $OLDPWD/gpurun_out/${name}_pmc${n}_$c.err )
        echo "pmc $c rc=$?" >> $LOG
      done
      prow=663000; [[ "$arg" =~ --rows[\ =]([0-9]+) ]] && prow=${BASH_REMATCH[1]}
      python scripts/pmc_traffic.py gpurun_out/${name}_pmc${n}_FETCH_SIZE gpurun_out/${name}_pmc${n}_WRITE_SIZE gpurun_out/${name}_k4_traffic$n.json $prow >> $LOG 2>&1
      rm -rf gpurun_out/${name}_pmc${n}_FETCH_SIZE gpurun_out/${name}_pmc${n}_WRITE_SIZE ;;
    pmctcc)   # only the L2 pass (requests, hits, misses, memory-side reads), e.g. pmctcc:--rows 5000000
      ( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum --output-format csv -d $OLDPWD/gpurun_out/${name}_pmctcc$n -o k -- python $OLDPWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-exact-kernel --no-end-to-end --no-side-runs $arg > /dev/null 2> $OLDPWD/gpurun_out/${name}_pmctcc$n.err )
      echo "-- pmc tcc pass rc=$?" >> $LOG
      python scripts/pmc_summary.py gpurun_out/${name}_pmctcc$n 2>&1 | grep -A8 "spgemm_topn_pruned" | head -24 >> $LOG
      rm -rf gpurun_out/${name}_pmctcc$n ;;
    envpmc)   # envpmc:VAR=value  -- the traffic passes with one environment variable set
      export ${arg}
      for c in FETCH_SIZE WRITE_SIZE; do
        ( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --pmc $c --output-format csv -d $OLDPWD/gpurun_out/${name}_pmc${n}_$c -o run -- python $OLDPWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-exact-kernel --no-end-to-end --no-side-runs > /dev/null 2> $OLDPWD/gpurun_out/${name}_pmc${n}_$c.err )
        echo "pmc $c ($arg) rc=$?" >> $LOG
      done
      python scripts/pmc_traffic.py gpurun_out/${name}_pmc${n}_FETCH_SIZE gpurun_out/${name}_pmc${n}_WRITE_SIZE gpurun_out/${name}_k4_traffic$n.json >> $LOG 2>&1
      rm -rf gpurun_out/${name}_pmc${n}_FETCH_SIZE gpurun_out/${name}_pmc${n}_WRITE_SIZE
      unset ${arg%%=*} ;;
    pmcsq1)   # only the pass the bench line's valu_issue_frac comes from
      ( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $OLDPWD/gpurun_out/${name}_pmcsq1 -o k -- python $OLDPWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-exact-kernel --no-end-to-end --no-side-runs $arg > /dev/null 2> $OLDPWD/gpurun_out/${name}_pmcsq1.err )
      echo "-- pmc pass rc=$?" >> $LOG
      python scripts/pmc_summary.py gpurun_out/${name}_pmcsq1 2>&1 | grep -A10 "spgemm_topn_pruned" | head -24 >> $LOG
      python scripts/pmc_counters.py gpurun_out/${name}_pmcsq1 gpurun_out/${name}_k4_counters.json >> $LOG 2>&1
      rm -rf gpurun_out/${name}_pmcsq1 ;;
    pmcsq)
      i=0
      for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
                  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVES" \
                  "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum"; do
        i=$((i+1))
        ( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --pmc $ctrs --output-format csv -d $OLDPWD/gpurun_out/${name}_pmcsq$i -o k -- python $OLDPWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-exact-kernel --no-end-to-end --no-side-runs $arg > /dev/null 2> $OLDPWD/gpurun_out/${name}_pmcsq$i.err )
        echo "-- pmc pass $i rc=$?: $ctrs" >> $LOG
        python scripts/pmc_summary.py gpurun_out/${name}_pmcsq$i 2>&1 | grep -A14 "spgemm_topn_pruned" | head -16 >> $LOG
        [ $i -eq 1 ] && python scripts/pmc_counters.py gpurun_out/${name}_pmcsq$i gpurun_out/${name}_k4_counters.json >> $LOG 2>&1
        rm -rf gpurun_out/${name}_pmcsq$i
      done ;;
    py)
      base=$(basename ${arg%% *} .py)
      timeout 1500 python $arg > gpurun_out/${name}_${base}$n.log 2>&1; echo "rc=$?" >> $LOG; tail -40 gpurun_out/${name}_${base}$n.log >> $LOG ;;
    sh)
      timeout 1500 bash -c "$arg" > gpurun_out/${name}_sh$n.log 2>&1; echo "rc=$?" >> $LOG; tail -15 gpurun_out/${name}_sh$n.log >> $LOG ;;
    *) echo "unknown step $step" >> $LOG ;;
  esac
done
cat $LOG
