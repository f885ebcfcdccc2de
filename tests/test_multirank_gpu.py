"""The N > 1 path on the REAL device ops with more than one rank: two and three processes on cuda:0,
process group on gloo with the host-staged transport of ``string_grouper_amd.distributed`` (RCCL refuses two ranks on one
device; the point is ``distributed.HipOps`` -- sg_selfjoin_range over group-position shares, sg_selfjoin_merge of ANOTHER
rank's pairs, sg_topn_expand_range, the scatter in ``gather_topn`` -- not the transport).  Each rank's rows and the gathered
result are compared bit for bit with ``oracle.port.sp_matmul_topn_port`` (string_grouper/string_grouper.py:733-752: the
block loop + vstack the ranks stand in for)."""
import os
import socket
import tempfile

import numpy as np
import pytest

from oracle import oracle as O
from oracle import port as P

pytestmark = pytest.mark.gpu

N_CPU = min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 4), 32)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _expected_selfjoin(workdir, key, names, top_n, thr, dtype):
    from tests._multirank_worker import save_expected
    (A,), _, _ = O.tfidf_sklearn(names, [names], dtype=dtype)
    save_expected(workdir, key, P.sp_matmul_topn_port(A, A.T, top_n, thr, True, N_CPU))


def _expected_match(workdir, key, master, dups, top_n, thr, dtype):
    from tests._multirank_worker import save_expected
    (A, B), _, _ = O.tfidf_sklearn(master + dups, [master, dups], dtype=dtype)
    save_expected(workdir, key, P.sp_matmul_topn_port(A, B.T, top_n, thr, True, N_CPU))


def _spawn(world, workdir, jobs, backend="gloo"):
    import torch.multiprocessing as mp
    from tests._multirank_worker import worker
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(worker, args=(world, _free_port(), workdir, jobs, ret, backend), nprocs=world, join=True)
    failures = []
    for r in range(world):
        assert r in ret, f"rank {r} returned nothing"
        for k, v in dict(ret[r]).items():
            if v:
                failures.append(f"rank {r} {k}: {v}")
    assert not failures, "\n".join(failures)
    return {r: dict(ret[r]) for r in range(world)}


@pytest.mark.timeout(1500)
def test_two_ranks_on_one_device_drive_the_real_device_ops():
    """World 2.  200 k names (>= 131 072: ``selfjoin_form_wanted`` without forcing; 16 % identical names: the library groups
    them): the self-join form over INTERLEAVED shares of the groups' positions, the same without groups, over contiguous
    ranges (both), the row-block form; fp64; master x duplicates top 20 / 0.7; a column of 8 distinct strings (fewer groups
    than top_n: ADVICE r03); the public API under enable_distributed()."""
    from string_grouper_amd.synth import synth_names
    from tests._multirank_worker import _names_of
    with tempfile.TemporaryDirectory(prefix="sg_mr_") as wd:
        big = {"n": 200_000, "seed": 1234}
        names = _names_of(big, synth_names)
        _expected_selfjoin(wd, "big_f32", names, 10, 0.8, np.float32)
        _expected_selfjoin(wd, "big_f64", names, 10, 0.8, np.float64)
        few = {"n": 20_000, "seed": 5, "distinct": 8}
        _expected_selfjoin(wd, "few_f32", _names_of(few, synth_names), 20, 0.8, np.float32)
        master = synth_names(60_000, 21)
        dups = synth_names(20_000, 22, perturb_of=master, perturb_frac=0.5)
        _expected_match(wd, "match_f32", master, dups, 20, 0.7, np.float32)
        sj = dict(kind="selfjoin", top_n=10, thr=0.8, expected="big_f32", **big)
        jobs = [
            dict(sj, tag="selfjoin_form_groups_interleaved", form="selfjoin", grouped=True),
            dict(sj, tag="selfjoin_form_rows_interleaved", form="selfjoin", grouped=False, options={"SG_COLLAPSE": "0"}),
            dict(sj, tag="selfjoin_form_groups_ranges", form="selfjoin", grouped=True, env={"SG_DIST_INTERLEAVE": "0"}),
            dict(sj, tag="selfjoin_form_rows_ranges", form="selfjoin", grouped=False, options={"SG_COLLAPSE": "0"},
                 env={"SG_DIST_INTERLEAVE": "0"}),
            dict(sj, tag="selfjoin_form_rows_in_row_order", form="selfjoin", grouped=False,
                 options={"SG_COLLAPSE": "0", "SG_PERMUTE": "0"}),
            dict(sj, tag="row_block_form", form="rowblock", env={"SG_DIST_SYM": "0"}),
            dict(sj, tag="selfjoin_form_f64", form="selfjoin", grouped=True, dtype="f64", expected="big_f64"),
            dict(kind="selfjoin", tag="fewer_groups_than_top_n", top_n=20, thr=0.8, expected="few_f32", form="selfjoin",
                 grouped=True, options={"SG_COLLAPSE": "1"}, env={"SG_DIST_SYM": "1"}, **few),
            dict(kind="match", tag="master_x_duplicates", top_n=20, thr=0.7, expected="match_f32", n_master=60_000, n_dups=20_000,
                 seed=21),
            dict(kind="api", tag="api", top_n=10, thr=0.8, n=30_000, seed=31, n_dups=6_000),
            dict(kind="api", tag="api_selfjoin_form", top_n=10, thr=0.8, n=30_000, seed=31, n_dups=6_000, env={"SG_DIST_SYM": "1"}),
            dict(kind="api", tag="api_f64", top_n=10, thr=0.8, n=12_000, seed=33, n_dups=3_000, dtype="f64"),
        ]
        got = _spawn(2, wd, jobs)
        assert len(got[0]) >= 40, got[0]


@pytest.mark.timeout(1500)
def test_three_ranks_with_uneven_shares():
    """World 3 on a size that 3 does not divide, with empty / short strings at the end of the column (the last rank's block):
    shares of 46 668 / 46 667 / 46 667 positions, blocks of uneven row counts; forced below the automatic threshold too."""
    from string_grouper_amd.synth import synth_names
    from tests._multirank_worker import _names_of
    with tempfile.TemporaryDirectory(prefix="sg_mr_") as wd:
        mid = {"n": 140_000, "seed": 77, "extra": ["", "AB", "ACME HOLDINGS INC"]}
        _expected_selfjoin(wd, "mid_f32", _names_of(mid, synth_names), 10, 0.8, np.float32)
        small = {"n": 25_001, "seed": 78, "extra": ["X"]}
        _expected_selfjoin(wd, "small_f32", _names_of(small, synth_names), 7, 0.75, np.float32)
        master = synth_names(30_001, 41)
        dups = synth_names(10_000, 42, perturb_of=master, perturb_frac=0.5)
        _expected_match(wd, "match3_f32", master, dups, 20, 0.7, np.float32)
        jobs = [
            dict(kind="selfjoin", tag="w3_groups_interleaved", top_n=10, thr=0.8, expected="mid_f32", form="selfjoin", grouped=True, **mid),
            dict(kind="selfjoin", tag="w3_rows_interleaved", top_n=10, thr=0.8, expected="mid_f32", form="selfjoin", grouped=False,
                 options={"SG_COLLAPSE": "0"}, **mid),
            dict(kind="selfjoin", tag="w3_groups_ranges", top_n=10, thr=0.8, expected="mid_f32", form="selfjoin", grouped=True,
                 env={"SG_DIST_INTERLEAVE": "0"}, **mid),
            dict(kind="selfjoin", tag="w3_row_block", top_n=10, thr=0.8, expected="mid_f32", form="rowblock", env={"SG_DIST_SYM": "0"}, **mid),
            dict(kind="selfjoin", tag="w3_small_forced", top_n=7, thr=0.75, expected="small_f32", form="selfjoin",
                 env={"SG_DIST_SYM": "1"}, **small),
            dict(kind="match", tag="w3_master_x_duplicates", top_n=20, thr=0.7, expected="match3_f32", n_master=30_001, n_dups=10_000,
                 seed=41),
            dict(kind="api", tag="w3_api", top_n=10, thr=0.8, n=20_001, seed=51, n_dups=4_001),
        ]
        _spawn(3, wd, jobs)


@pytest.mark.timeout(1500)
def test_eight_ranks_on_one_device_every_block_and_the_gathered_whole_equal_the_port():
    """World 8 -- the driver's largest configuration -- on the one GPU over gloo (the 8-rank bench
    line only compared a match count): 40 003 names (8 does not divide it; the self-join form is forced, as it would be
    taken from 131 072 names on), the self-join form over interleaved shares of the groups' positions and the row-block
    form; every rank's own rows and the gathered whole against the port, bit for bit; master x duplicates with blocks of
    uneven size."""
    from string_grouper_amd.synth import synth_names
    from tests._multirank_worker import _names_of
    with tempfile.TemporaryDirectory(prefix="sg_mr_") as wd:
        big = {"n": 40_000, "seed": 1234, "extra": ["", "AB", "ACME HOLDINGS INC"]}
        _expected_selfjoin(wd, "big_f32", _names_of(big, synth_names), 10, 0.8, np.float32)
        master = synth_names(20_003, 61)
        dups = synth_names(6_001, 62, perturb_of=master, perturb_frac=0.5)
        _expected_match(wd, "match8_f32", master, dups, 20, 0.7, np.float32)
        # ... and the form as it is CHOSEN: 140 000 names (>= 131 072), nothing forced, self-join form only
        auto = {"n": 140_000, "seed": 99, "extra": ["", "AB"]}
        _expected_selfjoin(wd, "auto_f32", _names_of(auto, synth_names), 10, 0.8, np.float32)
        jobs = [
            dict(kind="selfjoin", tag="w8_automatic_form_140k", top_n=10, thr=0.8, expected="auto_f32", form="selfjoin", grouped=True, **auto),
            dict(kind="selfjoin", tag="w8_groups_interleaved", top_n=10, thr=0.8, expected="big_f32", form="selfjoin", grouped=True,
                 env={"SG_DIST_SYM": "1"}, **big),
            dict(kind="selfjoin", tag="w8_row_block", top_n=10, thr=0.8, expected="big_f32", form="rowblock", env={"SG_DIST_SYM": "0"}, **big),
            dict(kind="match", tag="w8_master_x_duplicates", top_n=20, thr=0.7, expected="match8_f32", n_master=20_003, n_dups=6_001,
                 seed=61),
        ]
        got = _spawn(8, wd, jobs)
        assert all(len(got[r]) >= 11 for r in range(8)), got      # (3 + 3 + 3 + 2 checks per rank, all empty: _spawn has asserted that)
