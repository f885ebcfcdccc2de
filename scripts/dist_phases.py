"""Wall-clock of the phases of the multi-GPU driver on ONE rank (RCCL group of size 1): what the driver itself costs
beside the kernels.  python scripts/dist_phases.py [rows=663000]"""
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, ".")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")
from string_grouper_amd import _native as N  # noqa: E402
from string_grouper_amd import distributed as D  # noqa: E402
from string_grouper_amd.synth import synth_names  # noqa: E402
from string_grouper_amd.vectorizer import HipTfidfVectorizer  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 663000
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
ctx = N.Context()
ops = D.HipOps(ctx, lambda: HipTfidfVectorizer(dtype=np.float32, ctx=ctx))
names = synth_names(n, 1234)
block = HipTfidfVectorizer(dtype=np.float32, ctx=ctx).prepare(names)
os.environ["SG_DIST_SYM"] = "1"


def sync():
    ctx.sync()
    torch.cuda.synchronize()


def once(log):
    t = [time.perf_counter()]

    def tick(name):
        sync()
        t.append(time.perf_counter())
        log.setdefault(name, []).append((t[-1] - t[-2]) * 1e3)

    state, (A_local,) = D.sharded_tfidf(ops, [block])
    tick("sharded_tfidf (K1, all-reduce df, vocabulary, K2)")
    A_full = D.replicate_csr(ops, A_local)
    tick("replicate_csr (world 1: nothing)")
    post = ops.postings(A_full)
    tick("postings (K3)")
    n_index = ops.selfjoin_rows(A_full, post)          # (rows of the index: groups of identical rows)
    part = ops.selfjoin_range(A_full, post, 10, 0.8, 0, n_index)
    tick("selfjoin_range (K4p pass 1)")
    pairs = ops.selfjoin_pairs(part)
    sizes = [h[0] for h in D.all_headers([pairs.numel()], ops.device)]
    tick("header exchange")
    pairs_all = torch.cat(D.all_gather_ragged(pairs, None, sizes))
    tick("all-gather of the pairs")
    res = ops.selfjoin_merge(part, pairs_all, 0, n_index)
    tick("merge")
    res.free(); post.free(); A_full.free()
    tick("free")


log = {}
for rep in range(6):
    once(log if rep >= 2 else {})
tot = 0.0
for k, v in log.items():
    print(f"{k:55s} {np.mean(v):7.3f} ms")
    tot += np.mean(v)
print(f"{'sum':55s} {tot:7.3f} ms;  kernels (sg_stats): { {k: round(v, 3) for k, v in ctx.stats().items() if k.startswith('ms_')} }")
dist.destroy_process_group()
