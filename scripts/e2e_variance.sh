#!/bin/bash
# Round 6: match_strings() end to end a few times in fresh processes, with the SDMA engines (default) and with blit-kernel
# copies (HSA_ENABLE_SDMA=0): does the download of the match list explain the spread?   [E2E_ARGS=--no-exact-kernel] bash scripts/e2e_variance.sh [reps=3]
reps=${1:-3}
show() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); e = d['end_to_end']
        print('$1', 'step ms', round(d['ms_per_step'], 3), 'e2e f32', round(e['f32']['seconds'], 4), 'f64', round(e['f64']['seconds'], 4), 'list + download', e['f32']['split']['match_list_and_download_s'])"; }
for i in $(seq $reps); do
  HSA_ENABLE_SDMA=0 python bench.py --steps 20 --no-cpu-baseline --no-side-runs $E2E_ARGS 2>/dev/null | show "sdma off"
  python bench.py --steps 20 --no-cpu-baseline --no-side-runs $E2E_ARGS 2>/dev/null | show "default "
done
