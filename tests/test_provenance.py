"""The committed counter passes are quoted in the bench line only while the kernel sources are the ones they ran on
(string_grouper_amd/_provenance.py)."""
import json
import os
import shutil

from string_grouper_amd import _provenance as PV

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _copy_tree(tmp):
    for rel in PV.KERNEL_SOURCES:
        os.makedirs(os.path.dirname(os.path.join(tmp, rel)), exist_ok=True)
        shutil.copy(os.path.join(ROOT, rel), os.path.join(tmp, rel))
    os.makedirs(os.path.join(tmp, "profiles"))
    sha = PV.kernel_source_sha(tmp)
    json.dump({"workload_rows": 663000, "dtype": "f32", "kernel": "K4p-sym", "source_sha": sha,
               "traffic_bytes_per_launch_raw": 2.0e10, "source": "s", "note": "n"},
              open(os.path.join(tmp, "profiles", "k4_traffic.json"), "w"))
    json.dump({"workload_rows": 663000, "dtype": "f32", "kernel": "K4p-sym", "source_sha": sha,
               "per_launch": {"SQ_INSTS_VALU": 1.5e9}}, open(os.path.join(tmp, "profiles", "k4_counters.json"), "w"))
    return sha


def test_counters_are_quoted_for_the_sources_they_were_measured_on_and_dropped_after_an_edit(tmp_path):
    tmp = str(tmp_path)
    sha = _copy_tree(tmp)
    got = PV.committed_counters(tmp, 663000, "f32", "K4p-sym", 5.0)
    assert got["traffic"] == 2.0e10 and abs(got["valu_issue_frac"] - 1.5e9 * 4 / 1024 / 2.4e9 / 5e-3) < 1e-12
    assert sha[:12] in got["traffic_source"] and "traffic_stale" not in got and "valu_stale" not in got
    # another workload / kernel: nothing is quoted at all
    assert PV.committed_counters(tmp, 100000, "f32", "K4p-sym", 5.0) == {}
    assert PV.committed_counters(tmp, 663000, "f32", "K4p", 5.0) == {}
    # one byte of a kernel source changes: the numbers go, "stale" says why
    path = os.path.join(tmp, PV.KERNEL_SOURCES[0])
    with open(path, "ab") as f:
        f.write(b" ")
    got = PV.committed_counters(tmp, 663000, "f32", "K4p-sym", 5.0)
    assert got["traffic"] is None and "valu_issue_frac" not in got and "valu_insts_per_launch" not in got
    assert got["traffic_stale"]["status"] == "stale" and got["traffic_stale"]["measured_on_sources"] == sha[:12]
    assert got["valu_stale"]["current_sources"] == PV.kernel_source_sha(tmp)[:12] != sha[:12]


def test_files_without_a_recorded_sha_count_as_stale(tmp_path):
    tmp = str(tmp_path)
    _copy_tree(tmp)
    p = os.path.join(tmp, "profiles", "k4_traffic.json")
    d = json.load(open(p))
    del d["source_sha"]
    json.dump(d, open(p, "w"))
    got = PV.committed_counters(tmp, 663000, "f32", "K4p-sym", 5.0)
    assert got["traffic"] is None and got["traffic_stale"]["measured_on_sources"] == "unrecorded"
    assert "valu_issue_frac" in got


def test_the_committed_files_of_this_tree_either_match_the_sources_or_are_reported_stale():
    got = PV.committed_counters(ROOT, 663000, "f32", "K4p-sym", 5.0)
    assert ("traffic_stale" in got) != (got.get("traffic") is not None)


def test_a_pass_of_another_size_is_found_under_its_own_file_name(tmp_path):
    """Round 6: profiles/k4_traffic_5M.json (the 5 M self-join, past the Infinity Cache) next to the headline's file --
    the bench line of --rows 5000000 quotes it, the headline's line does not."""
    tmp = str(tmp_path)
    sha = _copy_tree(tmp)
    json.dump({"workload_rows": 5000000, "dtype": "f32", "kernel": "K4p-sym", "source_sha": sha,
               "traffic_bytes_per_launch_raw": 9.5e11, "source": "s", "note": "n"},
              open(os.path.join(tmp, "profiles", "k4_traffic_5M.json"), "w"))
    big = PV.committed_counters(tmp, 5000000, "f32", "K4p-sym", 147.0)
    assert big["traffic"] == 9.5e11 and "k4_traffic_5M.json" in big["traffic_source"] and "valu_issue_frac" not in big
    assert PV.committed_counters(tmp, 663000, "f32", "K4p-sym", 5.0)["traffic"] == 2.0e10
