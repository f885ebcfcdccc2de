"""The N > 1 path on the PRODUCT's transport: backend "nccl" (RCCL over xGMI), one rank per device.  Runs when the box shows
at least two GPUs and is skipped on a one-GPU box (where tests/test_multirank_gpu.py drives the same device ops with two,
three and eight ranks on one device over gloo).  A file of its own, sorted behind the others: a box with several GPUs is
one this repository's builder never had, so with `pytest -x` everything else has reported before this runs."""
import tempfile

import numpy as np
import pytest

from tests.test_multirank_gpu import _expected_match, _expected_selfjoin, _spawn

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(900)
def test_one_rank_per_device_over_rccl_when_the_box_has_more_than_one_gpu():
    """The product's transport: backend "nccl" (RCCL over xGMI), rank r on cuda:r, device tensors straight into the
    collectives -- on as many devices as the box shows (2 ... 8; skipped on a one-GPU box, where the gloo tests above
    stand in).  `distributed_self_join` in both forms, fp64, `distributed_match`, and the public API under
    enable_distributed(): every rank's block and the gathered whole against the CPU port, bit for bit (VERDICT r04, next
    2d: an 8-GPU driver box should produce parity evidence for the transport, not only a bench line)."""
    import torch
    world = min(torch.cuda.device_count(), 8)
    if world < 2:
        pytest.skip(f"{torch.cuda.device_count()} GPU visible: RCCL needs one device per rank")
    from string_grouper_amd.synth import synth_names
    from tests._multirank_worker import _names_of
    with tempfile.TemporaryDirectory(prefix="sg_mr_") as wd:
        big = {"n": 200_000, "seed": 1234, "extra": ["", "AB"]}
        names = _names_of(big, synth_names)
        _expected_selfjoin(wd, "big_f32", names, 10, 0.8, np.float32)
        _expected_selfjoin(wd, "big_f64", names, 20, 0.8, np.float64)
        master = synth_names(60_001, 21)
        dups = synth_names(20_000, 22, perturb_of=master, perturb_frac=0.5)
        _expected_match(wd, "match_f32", master, dups, 20, 0.7, np.float32)
        jobs = [
            dict(kind="selfjoin", tag="rccl_selfjoin_form", top_n=10, thr=0.8, expected="big_f32", form="selfjoin", grouped=True, **big),
            dict(kind="selfjoin", tag="rccl_row_block_form", top_n=10, thr=0.8, expected="big_f32", form="rowblock",
                 env={"SG_DIST_SYM": "0"}, **big),
            dict(kind="selfjoin", tag="rccl_selfjoin_form_f64_top20", top_n=20, thr=0.8, expected="big_f64", form="selfjoin", grouped=True,
                 dtype="f64", **big),
            dict(kind="match", tag="rccl_master_x_duplicates", top_n=20, thr=0.7, expected="match_f32", n_master=60_001, n_dups=20_000,
                 seed=21),
            dict(kind="api", tag="rccl_api", top_n=10, thr=0.8, n=30_000, seed=31, n_dups=6_000),
        ]
        got = _spawn(world, wd, jobs, backend="nccl")
        assert all(len(got[r]) >= 14 for r in range(world)), got
