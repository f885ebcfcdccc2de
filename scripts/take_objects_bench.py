"""Round 6: the frames' object gather (libsg_host.so, sg_host_gather_objects) on the two index patterns of a match list --
left side: sorted with runs, right side: near-random -- by thread count, against numpy's take; and, when a second build
lies beside it (string_grouper_amd/libsg_host_old.so), against that.   python scripts/take_objects_bench.py"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from string_grouper_amd.synth import synth_names  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = {}
for tag, name in (("current", "libsg_host.so"), ("old", "libsg_host_old.so")):
    path = os.path.join(ROOT, "string_grouper_amd", name)
    if os.path.exists(path):
        lib = C.PyDLL(path)
        lib.sg_host_gather_objects.restype = C.c_int
        lib.sg_host_gather_objects.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int]
        libs[tag] = lib
names = synth_names(663000, 1234)
vals = np.array(names, dtype=object)
rng = np.random.default_rng(0)
n = 2117764
left = np.sort(rng.integers(0, 663000, n))
right = rng.integers(0, 663000, n)
print("cpus in the affinity mask:", len(os.sched_getaffinity(0)))
for label, idx in (("left (sorted, runs)", left), ("right (random)", right)):
    t = time.perf_counter()
    ref = vals.take(idx)
    print(f"{label}: numpy take {1e3 * (time.perf_counter() - t):.1f} ms")
    for tag, lib in libs.items():
        row = []
        for T in (1, 4, 8, 16):
            best = 1e9
            for _ in range(4):
                out = np.empty(n, dtype=object)
                t = time.perf_counter()
                st = lib.sg_host_gather_objects(vals.ctypes.data, len(vals), idx.ctypes.data, n, out.ctypes.data, T)
                best = min(best, time.perf_counter() - t)
                assert st == 0 and out[0] is ref[0] and out[-1] is ref[-1]
                del out
            row.append(f"{T} threads {1e3 * best:6.1f} ms")
        print(f"   {tag:8s} " + " | ".join(row))
    del ref
