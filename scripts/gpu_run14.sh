#!/bin/bash
mkdir -p gpurun_out
export PYTHONPATH=$PWD
LOG=gpurun_out/run14.log
: > $LOG
make -s -C oracle
timeout 600 python -m pytest tests -m gpu -q -x -p no:cacheprovider --timeout=300 -k "fuzz or rccl" > gpurun_out/pytest14.log 2>&1
grep -E "passed|failed|Error" gpurun_out/pytest14.log | tail -5 >> $LOG
echo "== f64 bench" >> $LOG
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --dtype f64 > gpurun_out/bench14_f64.json 2> gpurun_out/bench14_f64.err
python -c "
import json; d=json.loads([l for l in open('gpurun_out/bench14_f64.json') if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['kernels_ms'], d['roofline']['frac'], d['roofline']['achieved'])" >> $LOG 2>&1
tail -3 gpurun_out/bench14_f64.err >> $LOG
echo "== f64 tile sweep" >> $LOG
python - >> $LOG 2>&1 <<'PY'
import os, time, numpy as np
from string_grouper_amd import _native as N
from string_grouper_amd.synth import synth_names
from string_grouper_amd.vectorizer import HipTfidfVectorizer
ctx = N.default_context(0)
names = synth_names(663000, 1234)
vec = HipTfidfVectorizer(dtype=np.float64, ctx=ctx); p = vec.prepare(names); vec.fit_prepared([p]); A = vec.transform_prepared(p)
for tile in (1024, 2048, 4096):
    post = ctx.postings_build(A, tile)
    for nb in (8,):
        os.environ["SG_DEPTH"] = str(nb)
        r = ctx.spgemm_topn(A, post, 10, 0.8, True); ctx.sync(); r.free()
        r = ctx.spgemm_topn(A, post, 10, 0.8, True); ctx.sync(); st = ctx.stats(); r.free()
        print("f64 tile", tile, "NB", nb, "%.1f ms" % st["ms_spgemm_topn"], "%.2f TB/s" % (st["spgemm_bytes"]/st["ms_spgemm_topn"]/1e9))
    post.free()
PY
cat $LOG
