show() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); e = d['end_to_end']
        print('$1', 'e2e f32', round(e['f32']['seconds'], 4), 'f64', round(e['f64']['seconds'], 4), 'list+download', e['f32']['split']['match_list_and_download_s'])"; }
python bench.py --no-cpu-baseline --no-side-runs --no-exact-kernel 2>/dev/null | show "flags  "
python bench.py --cpu-sample 2>/dev/null | show "full   "
python bench.py --no-cpu-baseline --no-side-runs --no-exact-kernel 2>/dev/null | show "flags  "
python bench.py --no-cpu-baseline 2>/dev/null | show "no-cpu "
python bench.py --no-cpu-baseline --no-side-runs 2>/dev/null | show "no-cpu no-side"
python bench.py --no-cpu-baseline --no-exact-kernel 2>/dev/null | show "no-cpu no-exact"
