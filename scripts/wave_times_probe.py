"""When do the waves of a share's launches end?  (a build with -DSG_DEBUG_WAVE_TIMES: scripts/build_variant.sh times
-DSG_DEBUG_WAVE_TIMES, run with SG_HIP_LIB=string_grouper_amd/libsg_hip_times.so)
python scripts/wave_times_probe.py [world=8] [rank=0] [rows=663000]"""
import ctypes
import sys

import numpy as np

sys.path.insert(0, ".")
from string_grouper_amd import _native as N  # noqa: E402
from string_grouper_amd import distributed as D  # noqa: E402
from string_grouper_amd.synth import synth_names  # noqa: E402
from string_grouper_amd.vectorizer import HipTfidfVectorizer  # noqa: E402

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rank = int(sys.argv[2]) if len(sys.argv) > 2 else 0
n = int(sys.argv[3]) if len(sys.argv) > 3 else 663000
ctx = N.Context()
vec = HipTfidfVectorizer(dtype=np.float32, ctx=ctx)
p = vec.prepare(synth_names(n, 1234))
vec.fit_prepared([p])
A = vec.transform_prepared(p)
post = ctx.postings_build(A)
n_index = ctx.postings_rows(post)[0]
lo, hi, step = (0, n_index, 1) if world == 1 else D.selfjoin_share(n_index, rank, world)
lib = N.lib()
buf = np.zeros(4 * 16384, np.uint64)
for rep in range(3):
    lib.sg_debug_read_wave_times(buf.ctypes.data_as(ctypes.c_void_p), 1)
    got = ctx.selfjoin_range(A, post, 10, 0.8, lo, hi, step)
    ctx.sync()
    ms = ctx.stats()["ms_spgemm_kernel"]
    got[0].free()
    ctx.device_free(got[1])
lib.sg_debug_read_wave_times(buf.ctypes.data_as(ctypes.c_void_p), 0)
w = buf.reshape(-1, 4)
print(f"world {world} rank {rank}, {n} names: kernel {ms:.3f} ms")
for name, part in (("rows", w[:8192]), ("parts", w[8192:])):
    part = part[part[:, 0] > 0]
    if not len(part):
        continue
    t0 = part[:, 0].min()
    start = (part[:, 0] - t0) / 100.0          # us
    end = (part[:, 1] - t0) / 100.0
    busy = end - start
    print(f"launch over {name}: {len(part)} waves; starts {np.percentile(start, [0, 50, 99, 100]).round(1)} us; "
          f"ends p1/p10/p50/p90/p99/max {np.percentile(end, [1, 10, 50, 90, 99, 100]).round(1)} us")
    print(f"    mean busy {busy.mean():.1f} us of the launch's {end.max():.1f}; the wave's slowest row p50/p90/p99/max "
          f"{np.percentile(part[:, 2] / 100.0, [50, 90, 99, 100]).round(1)} us")
    if name == "rows":
        late = np.argsort(-end)[:12]
        lens = np.diff(np.asarray(ctx.csr_indptr_host(A))) if hasattr(ctx, "csr_indptr_host") else None
        for wv in late:
            row = int(part[wv, 3] & 0xFFFFFFFF)
            print(f"      wave ends {end[wv]:8.1f} us; slowest row: position {row:7d} ({row / n_index:.2f} of the index) "
                  f"{part[wv, 2] / 100.0:7.1f} us, {int(part[wv, 3] >> 32)} pairs scored")
    # how many waves are still at work over the launch
    edges = np.linspace(0, end.max(), 11)
    alive = [(int(((start <= e) & (end > e)).sum())) for e in edges[:-1]]
    print("    waves at work at 0, 10, ... 90 % of the launch:", alive)
