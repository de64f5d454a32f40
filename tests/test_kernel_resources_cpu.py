"""What the compiler made of the FAST kernels, checked without a GPU (hipcc cross-compiles gfx950): the register budgets the launch
bounds promise, no spills in the small-scene loops, and no FLAT memory instruction anywhere in the translation unit.

Why a test: both have regressed silently before. A `volatile` access through a generic pointer compiles to `flat_load` /
`flat_store` with system scope and a `vmcnt(0)` wait (the compiler does not infer the LDS address space for volatile accesses) --
the large-scene kernels' list walk carried five of those per round for most of round 4 -- and one more scalar value live across the
small-scene loop spills five registers (-1.4 %, profiles/HISTORY.md section 4.2). The command is the Makefile's own (`make -n`), with the
assembly and the resource remarks asked for instead of an object file."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "kajo_amd", "csrc")


def _compile(unit):
    if shutil.which("hipcc") is None or shutil.which("make") is None:
        pytest.skip("hipcc / make not available")
    obj = os.path.join(CSRC, "build", "kernel_%s.o" % unit)
    plan = subprocess.run(["make", "-n", "-B", "-C", CSRC, obj], capture_output=True, text=True, check=True).stdout
    cmd = next(l for l in plan.splitlines() if l.startswith("hipcc") and "kernel_%s.hip" % unit in l).split()
    tmp = tempfile.mkdtemp(prefix="kajo_res_")
    asm = os.path.join(tmp, "k.s")
    i = cmd.index("-c")
    cmd = cmd[:i] + ["-S", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage"] + cmd[i + 1:]
    cmd[cmd.index("-o") + 1] = asm
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    res, name = {}, None
    for line in r.stderr.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            res[name] = {}
            continue
        m = re.search(r"remark:\s+(VGPRs|SGPRs Spill|VGPRs Spill|Occupancy \[waves/SIMD\]|ScratchSize \[bytes/lane\]): (\d+)", line)
        if m and name:
            res[name][m.group(1).split(" [")[0]] = int(m.group(2))
    text = open(asm).read()
    shutil.rmtree(tmp, ignore_errors=True)
    return res, text


def test_fast_kernels_keep_their_register_budgets_and_use_no_flat_accesses():
    res, asm = _compile("fast")
    # the two small-scene instances and the small-frame kernel: five waves per SIMD (96 VGPRs), nothing spilled
    for k in ("kajo_render_fast", "kajo_render_fast_lights", "kajo_render_fast_split"):
        assert k in res, sorted(res)
        r = res[k]
        assert r["Occupancy"] == 5 and r["VGPRs"] <= 96, (k, r)
        assert r["VGPRs Spill"] == 0 and r["SGPRs Spill"] == 0 and r["ScratchSize"] == 0, (k, r)
    # the large-scene kernels, an instance per home of the grid's cell lists (LDS: _lg): four waves per SIMD, no vector spills (one spilled
    # VGPR cost the list kernel 2.4 GB of scratch traffic per launch in round 5); scalar spills 65 -> 44-46 in the list kernels (round 6)
    for k in ("kajo_render_fast_big", "kajo_render_fast_biglist", "kajo_render_fast_big_lg", "kajo_render_fast_biglist_lg"):
        r = res[k]
        assert r["Occupancy"] == 4 and r["VGPRs"] <= 116 and r["VGPRs Spill"] == 0 and r["ScratchSize"] == 0 and r["SGPRs Spill"] <= 50, (k, r)
    # every pointer of these kernels has a known home (LDS or global): a flat access is an address space the compiler could not infer
    flat = [l.strip() for l in asm.splitlines() if re.match(r"\s+flat_(load|store|atomic)", l)]
    assert not flat, flat[:5]
    # ... and the cross-lane words of the list walk are LDS instructions, not scratch or global ones
    body = asm[asm.index("kajo_render_fast_biglist:"):]
    body = body[:body.index("s_endpgm")]
    assert "ds_bpermute_b32" in body and "ds_write_b32" in body


def test_strict_kernels_keep_their_register_budgets():
    res, asm = _compile("strict")
    # small scenes: four waves per SIMD, no vector spills (the scalar ones are loop-invariant values parked in a VGPR's lanes)
    for k in ("kajo_render_strict", "kajo_render_strict_lights", "kajo_render_strict_split"):
        assert k in res, sorted(res)
        r = res[k]
        assert r["Occupancy"] == 4 and r["VGPRs"] <= 128 and r["VGPRs Spill"] == 0 and r["ScratchSize"] == 0, (k, r)
        body = asm[asm.index(k + ":"):]
        body = body[:body.index("s_endpgm")]
        assert not re.search(r"\n\s+flat_(load|store|atomic)", body), k
    # large scenes: an instance per home of the grid's cell lists (LDS: _lg), typed loads in each
    for k in ("kajo_render_strict_big", "kajo_render_strict_biglist", "kajo_render_strict_big_lg", "kajo_render_strict_biglist_lg"):
        r = res[k]
        assert r["Occupancy"] == 4 and r["VGPRs"] <= 128 and r["VGPRs Spill"] <= 16, (k, r)
        body = asm[asm.index(k + ":"):]
        body = body[:body.index("s_endpgm")]
        assert not re.search(r"\n\s+flat_(load|store|atomic)", body), k
    # (the known-answer kernels walk the grid through a pointer of either home: the only flat accesses of the unit)
    for k in ("kajo_kat_shade_strict", "kajo_kat_trace_strict"):
        body = asm[asm.index(k + ":"):]
        body = body[:body.index("s_endpgm")]
        assert sorted(re.findall(r"\n\s+(flat_\w+)", body)) == ["flat_load_dwordx2", "flat_load_ushort"], k


def _body(asm, kernel):
    body = asm[asm.index(kernel + ":"):]
    return body[:body.index("s_endpgm")]


def _valu(asm, kernel):
    return len(re.findall(r"\n\s+v_\w+", _body(asm, kernel)))


def _scratch_in_loops(asm, kernel):
    """scratch_* instructions of the kernel that sit in a basic block of a loop (the compiler's block comments say which are)."""
    hits, in_loop = [], False
    for line in _body(asm, kernel).splitlines():
        t = line.strip()
        if t.startswith(".LBB") or t.startswith("; %bb."):
            in_loop = "Loop" in t
        elif t.startswith(";") and "Loop" in t:
            in_loop = True  # (continuation lines of a block's comment: "Parent Loop", "Inner Loop Header")
        elif in_loop and t.startswith("scratch_"):
            hits.append(t)
    return hits


def test_exact_kernels_keep_their_register_budgets():
    """The EXACT build (round 5): STRICT's decision arithmetic in FAST's register budget. The one-light instance -- what bench.py times --
    runs five waves per SIMD with nothing spilled to scratch (+8 % over four waves, profiles/r05_notes.txt); the any-number-of-lights
    instance carries the inline shadow walk and spills a few registers at that occupancy (measured: still the faster schedule)."""
    res, asm = _compile("exact")
    r = res["kajo_render_exact"]
    assert r["Occupancy"] == 5 and r["VGPRs"] <= 96 and r["VGPRs Spill"] == 0 and r["ScratchSize"] == 0 and r["SGPRs Spill"] <= 10, r
    # Round 6: the instance of any number of lights (10 VGPRs / 28 bytes of scratch in round 5: +14 MB of HBM traffic per launch of the
    # caustics scene, profiles/r06_notes.txt) and the small-frame kernel spill no vector register either -- what they parked were
    # loop-invariant addresses (a lane's LDS word, its buffer slot as a 64-bit pair, threadIdx.x), now formed again where they are used.
    for k in ("kajo_render_exact_split", "kajo_render_exact_lights"):
        r = res[k]
        assert r["Occupancy"] == 5 and r["VGPRs"] <= 96 and r["VGPRs Spill"] == 0 and r["ScratchSize"] == 0 and r["SGPRs Spill"] <= 16, (k, r)
        assert not _scratch_in_loops(asm, k)
    # large scenes: no vector spills; the scalar ones were 76 / 67 in the list kernels before the instances of scenes with visibility lists
    # dropped the code of general sphere records, scaled planes and the every-sphere fallback (round 6)
    for k, sgpr in (("kajo_render_exact_big", 44), ("kajo_render_exact_biglist", 62), ("kajo_render_exact_big_lg", 40), ("kajo_render_exact_biglist_lg", 60)):
        r = res[k]
        assert r["Occupancy"] == 4 and r["VGPRs"] <= 128 and r["VGPRs Spill"] == 0 and r["SGPRs Spill"] <= sgpr, (k, r)
    for k in res:
        if k.startswith("kajo_render_exact"):
            assert not re.search(r"\n\s+flat_(load|store|atomic)", _body(asm, k)), k
    # the IEEE quotient and root are formed by hand (no range scaling): hipcc's own sequences would show up as these
    body = _body(asm, "kajo_render_exact")
    assert "v_div_scale_f32" not in body and "v_div_fmas_f32" not in body and "v_div_fixup_f32" in body


# Static VALU instruction counts of the small-scene kernels (the whole kernel, loops counted once), from the compiler's assembly of the
# Makefile's own command. The loop's time follows its VALU instruction count (STRICT / FAST: 2.28 x the instructions, 2.26 x the time,
# profiles/HISTORY.md section 4.3), and round 4 lost 2-3 % twice to changes that "only" added instructions to it (commit 8218d68). A change that
# adds more than ~3 % has to raise its budget here, knowingly. Measured at the time of writing: 1396, 1360, 2227, 2471, 2766, 3003
# (the STRICT / EXACT kernels carry two sphere loops and two plane loops since round 5 -- the ones for scenes of (centre, radius) spheres
# and rigid planes, +7 % / +6.5 %, and the general ones: ~170-250 instructions more in the text, fewer executed).
# Round 6: 1456, 1420, 2278, 2520, 2768, 3004 -- the FAST / EXACT kernels' prologue forms a parted workgroup's compact side-buffer slot (one
# integer division) and reads a group carried over from the launch before, their epilogue writes it (+60 instructions outside the loop;
# the loop's blocks count 168 / 353 / 244 / 286 as before, tools/isa_blocks.sh).
VALU_BUDGET = {
    ("fast", "kajo_render_fast"): 1500,
    ("fast", "kajo_render_fast_lights"): 1465,
    ("exact", "kajo_render_exact"): 2290,
    ("exact", "kajo_render_exact_lights"): 2540,
    ("strict", "kajo_render_strict"): 2850,
    ("strict", "kajo_render_strict_lights"): 3090,
}


@pytest.mark.parametrize("unit", ["fast", "exact", "strict"])
def test_instruction_counts_of_the_small_scene_loops(unit):
    _, asm = _compile(unit)
    for (u, kernel), budget in VALU_BUDGET.items():
        if u != unit:
            continue
        n = _valu(asm, kernel)
        assert n <= budget, "%s: %d VALU instructions, budget %d" % (kernel, n, budget)
        assert n >= 0.85 * budget, "%s: %d VALU instructions -- far below the budget %d: lower it to keep the guard tight" % (kernel, n, budget)
