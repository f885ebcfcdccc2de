"""CPU guard on the code hipcc generates for the pruned multiply's tile loop (no GPU needed: hipcc cross-compiles gfx950).

The kernel (string_grouper_amd/csrc/sg_spgemm_pruned.hip) is VALU-issue bound and sits at the 128-VGPR limit of four
waves per SIMD; every round of work on it has shown the same failure: an innocent edit makes the register allocator
spill something INSIDE the tile loop -- typically a prefetched batch, stored to scratch right behind its load, which
turns the loop's `s_waitcnt vmcnt(3..5)` into `vmcnt(0)` and costs 10-40 % on the GPU (DESIGN.md section 4,
profiles/r02_sessionAH_AI_*, r02_sessionAL_*).  None of that is visible without a GPU run -- except in the ISA.  This
test compiles the file to assembly and checks, for the four kernels that carry the headline paths (f32 / f64, self-join
form / one-sided, 4096-column tile):
  * the fast path of the tile loop (the block with four LDS adds and four re-zeroing stores) touches no scratch memory
    and keeps its size;
  * no value is spilled right behind its load anywhere in the tile loop (`s_waitcnt vmcnt(0)` + `scratch_store`);
  * the tile loop still waits with a COUNT for its prefetched batches (vmcnt(3) or more somewhere in it);
  * spills stay at the few the row set-up has always had.
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "string_grouper_amd", "csrc", "sg_spgemm_pruned.hip")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

# template arguments <T, TILE_LOG2, SYM, WIDE, FOLD_LOG2> -> mangled fragment (FOLD_LOG2 = 0: the tile-by-tile form)
KERNELS = {
    "f32 self-join": "IfLi12ELb1ELb0ELi0ELb0EE",
    "f32 one-sided": "IfLi12ELb0ELb0ELi0ELb0EE",
    "f64 self-join": "IdLi12ELb1ELb0ELi0ELb0EE",
    "f64 one-sided": "IdLi12ELb0ELb0ELi0ELb0EE",
}
# the stream form (FOLD_LOG2 = 3): rounds of four adds, no per-posting re-zeroing
STREAM_KERNELS = {
    "f32 self-join": "IfLi12ELb1ELb0ELi3ELb0EE",
    "f32 one-sided": "IfLi12ELb0ELb0ELi3ELb0EE",
    "f64 self-join": "IdLi12ELb1ELb0ELi3ELb0EE",
    "f64 one-sided": "IdLi12ELb0ELb0ELi3ELb0EE",
}
# ... and the second launch over rows of 65 .. 128 non-zeros (WIDE): same loop, checked for its hand-written loads
STREAM_WIDE_KERNELS = {
    "f32 self-join wide": "IfLi12ELb1ELb1ELi3ELb0EE",
    "f32 one-sided wide": "IfLi12ELb0ELb1ELi3ELb0EE",
    "f64 self-join wide": "IdLi12ELb1ELb1ELi3ELb0EE",
    "f64 one-sided wide": "IdLi12ELb0ELb1ELi3ELb0EE",
}
# ... and the instantiation for a rank's share of the rows and for the rows in parts (SHARE: the multi-GPU form)
STREAM_SHARE_KERNELS = {
    "f32 self-join share": "IfLi12ELb1ELb0ELi3ELb1EE",
    "f64 self-join share": "IdLi12ELb1ELb0ELi3ELb1EE",
}
# vgpr spills allowed (all outside the trip -- the test below checks that no scratch access lies inside it).  f64 -- the
# reference's DEFAULT dtype -- was built for three waves per SIMD through round 3 (a spill inside its round loop cost
# 14.9 -> 16.4 ms in round 2, profiles/r02_sessionAL_f64_bisect.log); round 4's loop fits four (7.0 -> 6.5 ms at 663 k).
# (round 5: the f64 self-join kernels spill 12 / 10 around the row's set-up -- the row's own diagonal is summed there and the
#  survivor routine, whose registers the caller must leave alone, grew by the second filter's four bucket reads; still none
#  inside the trip)
STREAM_LIMITS = {"f32 self-join": 16, "f32 one-sided": 16, "f64 self-join": 12, "f64 one-sided": 8,
                 "f32 self-join share": 16, "f64 self-join share": 12}
STREAM_VGPRS = {"f32 self-join": 128, "f32 one-sided": 128, "f64 self-join": 128, "f64 one-sided": 128,   # 4 waves per SIMD
                "f32 self-join share": 128, "f64 self-join share": 128}
# (vgpr spills allowed, instructions of the fast-path block allowed)
LIMITS = {"f32 self-join": (16, 95), "f32 one-sided": (16, 95), "f64 self-join": (24, 95), "f64 one-sided": (24, 95)}


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    if not (os.path.exists(HIPCC) or shutil.which(HIPCC)):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("isa") / "pruned.s"
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-function",
           "-S", "--cuda-device-only", "-o", str(out), SRC]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return out.read_text()


def kernel_body(asm, frag):
    m = re.search(r"^(_Z25spgemm_topn_pruned_kernel%s\w*):" % frag, asm, re.M)
    assert m, frag
    start = m.start()
    end = asm.index("s_endpgm", start)
    return m.group(1), asm[start:end]


def blocks_of(body):
    """[(label, [instruction lines], in_tile_loop)]; the tile loop = the blocks annotated with loop depth >= 3 whose
    header is the loop that holds the fast path (found from the fast path itself)."""
    out, cur = [], None
    for line in body.split("\n"):
        m = re.match(r"^(\.LBB\S+):(.*)$", line)
        if m:
            cur = {"label": m.group(1), "ins": [], "raw": [], "note": m.group(2)}
            out.append(cur)
            continue
        s = line.strip()
        if cur is None or not s or s.startswith(";") or s.startswith("."):
            if cur is not None and "Loop" in s and not cur["note"]:
                cur["note"] = s
            continue
        cur["ins"].append(s.split(";")[0].strip())
        cur["raw"].append(s)
        if ";" in s and "Loop" in s and not cur["note"]:
            cur["note"] = s
    return out


def is_fast_path(b):
    adds = [i for i, x in enumerate(b["ins"]) if x.startswith("ds_add_rtn_u32")]
    zeros = [i for i, x in enumerate(b["ins"]) if x.startswith("ds_write_b32")]
    return len(adds) == 4 and len(zeros) >= 4


@pytest.mark.timeout(1200)
@pytest.mark.parametrize("which", list(KERNELS))
def test_tile_loop_of_the_pruned_kernel_keeps_its_shape(asm, which):
    name, body = kernel_body(asm, KERNELS[which])
    meta = asm[asm.index(".name:           " + name):]
    spills = int(re.search(r"\.vgpr_spill_count:\s*(\d+)", meta).group(1))
    vgprs = int(re.search(r"\.vgpr_count:\s*(\d+)", meta).group(1))
    max_spills, max_fast = LIMITS[which]
    assert vgprs <= 128, (which, vgprs)                      # four waves per SIMD
    assert spills <= max_spills, (which, spills)

    blocks = blocks_of(body)
    fast = [b for b in blocks if is_fast_path(b)]
    assert len(fast) == 4, (which, len(fast))                # the loop is unrolled four tiles deep
    for b in fast:
        assert not any(x.startswith("scratch_") for x in b["ins"]), (which, b["label"], "scratch access in the fast path")
        assert len(b["ins"]) <= max_fast, (which, b["label"], len(b["ins"]))
        valu = sum(1 for x in b["ins"] if x.startswith("v_"))
        assert valu <= 66, (which, b["label"], valu)

    # the tile loop: everything between the first and the last fast-path block
    i0, i1 = blocks.index(fast[0]), blocks.index(fast[-1])
    loop = blocks[max(0, i0 - 12): i1 + 1]
    flat = [x for b in loop for x in b["ins"]]
    for k, a in enumerate(flat):
        if a.startswith("s_waitcnt vmcnt(0)"):
            assert not any(b.startswith("scratch_store") for b in flat[k + 1:k + 4]), \
                (which, "a value is spilled right behind its load inside the tile loop")
    counts = [int(m.group(1)) for x in flat for m in [re.match(r"s_waitcnt vmcnt\((\d+)\)", x)] if m]
    assert max(counts) >= 3, (which, counts)                 # the prefetched batches are waited for by count


def is_stream_round(b):
    adds = [i for i, x in enumerate(b["ins"]) if x.startswith("ds_add_rtn_u32")]
    return len(adds) == 4


@pytest.mark.timeout(1200)
@pytest.mark.parametrize("which", list(STREAM_KERNELS) + list(STREAM_SHARE_KERNELS))
def test_round_loop_of_the_stream_form_keeps_its_shape(asm, which):
    """The stream form's loop: four unrolled rounds, each one block with the four LDS adds of a round and no scratch
    access; the prefetched rounds are waited for by count; spills stay out of the loop."""
    name, body = kernel_body(asm, {**STREAM_KERNELS, **STREAM_SHARE_KERNELS}[which])
    meta = asm[asm.index(".name:           " + name):]
    spills = int(re.search(r"\.vgpr_spill_count:\s*(\d+)", meta).group(1))
    vgprs = int(re.search(r"\.vgpr_count:\s*(\d+)", meta).group(1))
    assert vgprs <= STREAM_VGPRS[which], (which, vgprs)
    assert spills <= STREAM_LIMITS[which], (which, spills)
    blocks = blocks_of(body)
    rounds = [b for b in blocks if is_stream_round(b)]
    assert len(rounds) == 4, (which, len(rounds))
    for b in rounds:
        assert not any(x.startswith("scratch_") for x in b["ins"]), (which, b["label"], "scratch access in a round")
        valu = sum(1 for x in b["ins"] if x.startswith("v_"))
        assert valu <= 80, (which, b["label"], valu)
    # no spill or reload between the first and the last round of the trip (scratch accesses are vector memory
    # operations the compiler waits for with vmcnt(0): they would drain the rounds in flight)
    i0, i1 = blocks.index(rounds[0]), blocks.index(rounds[-1])
    for b in blocks[i0:i1 + 1]:
        assert not any(x.startswith("scratch_") for x in b["ins"]), (which, b["label"], "scratch access inside the trip")
        # ... and no wait of the compiler's own for ALL loads: the only vmcnt(0) inside the trip are the hand-written ones in
        # front of the calls that score survivors (round 3 had one behind every load of the next visits' segment ends --
        # a full memory round trip with the rounds in flight drained, every fourth visit; round 4 loads the ends by hand)
        for x in b["raw"]:
            if x.startswith("s_waitcnt") and "vmcnt(0)" in x:
                assert "; rounds" in x, (which, b["label"], x)


def _regs_of(text):
    """VGPR numbers an instruction line mentions (v7, v[8:11])."""
    out = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", text):
        out.update(range(int(a), int(b) + 1))
    for a in re.findall(r"\bv(\d+)\b", text):
        out.add(int(a))
    return out


@pytest.mark.timeout(1200)
@pytest.mark.parametrize("which", list(STREAM_KERNELS) + list(STREAM_WIDE_KERNELS) + list(STREAM_SHARE_KERNELS))
def test_nothing_reads_a_round_between_its_hand_written_load_and_its_wait(asm, which):
    """The stream form loads its rounds with `global_load_dwordx4` in inline assembly and waits for them with a hand-counted
    `s_waitcnt vmcnt(3)`: the compiler does not know that these registers are in flight, so a copy or a spill of one of
    them between the load and the wait would read garbage -- silently.  Every such load must be followed (in layout
    order) by its own wait before anything else names its registers, and there must be exactly four round variables."""
    name, body = kernel_body(asm, {**STREAM_KERNELS, **STREAM_WIDE_KERNELS, **STREAM_SHARE_KERNELS}[which])
    raw = body.split("\n")
    loads = []
    for i, x in enumerate(raw):
        if x.strip().startswith("global_load_dwordx4") and i > 0 and "#ASMSTART" in raw[i - 1]:
            m = re.match(r"global_load_dwordx4 v\[(\d+):(\d+)\]", x.strip())
            assert m, x
            loads.append((i, int(m.group(1)), int(m.group(2))))
    assert len(loads) >= 8, (which, len(loads))            # four in the prologue, four in the loop
    assert len({(a, b) for _, a, b in loads}) == 4, (which, sorted({(a, b) for _, a, b in loads}))
    for i, a, b in loads:
        regs = set(range(a, b + 1))
        want = "s_waitcnt vmcnt(3) ; round v[%d:%d]" % (a, b)
        for j in list(range(i + 1, len(raw))) + list(range(0, i)):   # (the loop wraps: a round loaded at its end is waited for at its head)
            t = raw[j].strip()
            if t.startswith(want) or t.startswith("s_waitcnt vmcnt(0) ; rounds"):
                break
            if not t or t.startswith(";") or t.startswith(".") or t.startswith("s_waitcnt vmcnt(3) ; round") or \
                    t.startswith("s_endpgm"):
                continue
            ins = t.split(";")[0]
            if ins.startswith("global_load_dwordx4") and "#ASMSTART" in raw[j - 1]:
                assert not (_regs_of(ins.split(",")[0]) & regs), (which, j, t)
                touched = _regs_of(",".join(ins.split(",")[1:]))
            else:
                touched = _regs_of(ins)
            assert not (touched & regs), (which, "line %d names v[%d:%d] before its wait: %s" % (j, a, b, t))
        else:
            raise AssertionError((which, "no wait found for the load at line %d" % i))
