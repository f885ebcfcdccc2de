#!/bin/bash
mkdir -p gpurun_out
export PYTHONPATH=$PWD
LOG=gpurun_out/run15.log
: > $LOG
make -s -C oracle
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider --timeout=300 -k "fused or golden" > gpurun_out/pytest15.log 2>&1
grep -E "passed|failed|Error|assert" gpurun_out/pytest15.log | tail -8 >> $LOG
echo "== bench end-to-end" >> $LOG
true > gpurun_out/bench15.json 2> gpurun_out/bench15.err
python -c "
import json; d=json.load(open('gpurun_out/bench15.json')); print(d['value'], d['ms_per_step'], d['kernels_ms'], d['roofline']['frac'], d.get('end_to_end_match_strings_s'), d.get('end_to_end_rows'), d['cpu_baseline']['value'], d['parity_on_sample'], d['roofline'].get('traffic'))" >> $LOG 2>&1
tail -3 gpurun_out/bench15.err >> $LOG
cat $LOG
