"""The frame is a function of (scene, parameters, passes done) alone -- in EVERY numerics build (round 6).

The reference adds `radiance / m_samples` to a pixel's total pass by pass in one fixed order (renderer/cpu/Renderer.cpp:66-71); STRICT does
exactly that. FAST and EXACT (small scenes) add the passes in groups of four by their ABSOLUTE numbers -- passes 1-4, 5-8, ...: a group is
summed from zero in pass order, the group sums are added to the total in group order, a group in progress is added last
(include/kajo_hip.h kajo_hip_render; integrator.inc.hip GROUPS) -- so that the work of a launch can be divided among waves, workgroups
(the parted launch tail) and GPUs. Round 5 sized the groups from the LAUNCH (a quarter of it), which made the last bits of the default
plugin run depend on how `hip::Scheduler` happened to batch the passes (its batch size comes from a wall clock). Here: any cut of the same
passes into render calls gives one buffer, bit for bit; through the C ABI, the parted tail, the small-frame kernels, several tile owners
and `kajo_render --batch`."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from kajo_amd import capi
from kajo_amd.renderer import HipRenderer

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "kajo_amd", "host", "kajo_render")
SEED = 0o715517
CUTS = [(16,), (8, 8), (4, 12), (1, 2, 13), (3, 5, 8)]


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def render_cuts(sc, w, h, cuts, owners=1, warm=0, **kw):
    """The frame after sum(cuts) passes rendered as len(cuts) render calls, composed from `owners` tile owners. warm: passes rendered (and
    discarded by a reset) first, so that the handle has measured its launch order and parts the tail of the launches that allow it."""
    import torch
    parts = [HipRenderer(sc, w, h, seed=SEED, passes_per_launch=16, tile_index=i, tile_count=owners, **kw) for i in range(owners)]
    try:
        tail = 0
        for p in parts:
            if warm:
                p.render(warm).wait()
                p.reset()
            for c in cuts:
                p.render(c).wait()
                tail += p.counters()["tailGroups"]
        if owners == 1:
            return parts[0].radiance()[..., :3].copy(), tail
        nbytes = parts[0].tile_buffer()[1]
        gathered = torch.empty(owners * nbytes // 4, dtype=torch.float32, device="cuda")
        hip = C.CDLL("libamdhip64.so")
        for i, p in enumerate(parts):
            ptr, _ = p.tile_buffer()
            assert hip.hipMemcpy(C.c_void_p(gathered.data_ptr() + i * nbytes), C.c_void_p(ptr), C.c_size_t(nbytes), C.c_int(3)) == 0
        parts[0].compose(gathered.data_ptr())
        return parts[0].radiance()[..., :3].copy(), tail
    finally:
        for p in parts:
            p.close()


@pytest.mark.parametrize("mode", ["fast", "exact"])
@pytest.mark.parametrize("owners", [1, 3])
def test_any_cut_of_the_same_passes_gives_one_buffer(scenes, mode, owners):
    """1280 x 720: the unsplit kernels, with the launch tail in parts wherever a launch is whole groups (the handles are warmed first)."""
    sc = scenes["spheres_a169"]
    kw = dict(exact=True) if mode == "exact" else {}
    want, tail = render_cuts(sc, 1280, 720, CUTS[0], owners=owners, warm=1, **kw)
    assert tail > 0 or owners == 3  # (a third of 14 400 blocks is one round of the wave slots: not parted)
    assert np.isfinite(want).all(-1).mean() > 0.999
    for cuts in CUTS[1:]:
        got, _ = render_cuts(sc, 1280, 720, cuts, owners=owners, warm=1, **kw)
        assert np.array_equal(bits(got), bits(want)), (mode, owners, cuts)
    # ... and neither does the division of a launch: no parts, image order, one launch per pass
    for flags, ppl in ((capi.KAJO_FLAG_NO_SPLIT, 16), (capi.KAJO_FLAG_NO_REORDER, 16)):
        got, tail = render_cuts(sc, 1280, 720, (16,), owners=owners, warm=1, flags=flags, **kw)
        assert tail == 0
        assert np.array_equal(bits(got), bits(want)), (mode, owners, flags)


@pytest.mark.parametrize("mode", ["fast", "exact"])
def test_small_frames_and_one_pass_launches(scenes, mode):
    """256 x 144 runs the SPLIT kernels (waves of a block divide the passes or the samples; wave 0 forms the sums); 1 x 16 launches of
    one pass carry every group across four launches."""
    sc = scenes["spheres_a169"]
    kw = dict(exact=True) if mode == "exact" else {}
    want, _ = render_cuts(sc, 256, 144, (16,), **kw)
    for cuts in CUTS[1:] + [(1,) * 16, (5, 11), (15, 1)]:
        got, _ = render_cuts(sc, 256, 144, cuts, **kw)
        assert np.array_equal(bits(got), bits(want)), (mode, cuts)
    got, _ = render_cuts(sc, 256, 144, (16,), flags=capi.KAJO_FLAG_NO_SPLIT, **kw)
    assert np.array_equal(bits(got), bits(want))


def test_the_groups_are_the_documented_sums(scenes):
    """The EXACT frame of 10 passes against a host model of include/kajo_hip.h's rule built from the ten passes' own terms (each rendered
    alone into a zeroed buffer): ((0 + (t1 + t2 + t3 + t4)) + (t5 + ... + t8)) + (t9 + t10), every sum from zero in pass order."""
    sc = scenes["spheres_a169"]
    W, H = 320, 180
    with HipRenderer(sc, W, H, seed=SEED, exact=True) as r:
        terms = []
        for p in range(1, 11):
            r.reset()
            r.set_pass_count(p - 1)
            terms.append(r.render(1).radiance()[..., :3].copy())
        r.reset()
        got = r.render(10).radiance()[..., :3].copy()
    total = np.zeros_like(terms[0])
    for g in range(0, 10, 4):
        group = np.zeros_like(total)
        for t in terms[g:g + 4]:
            group = group + t
        total = total + group
    assert total.dtype == np.float32
    assert np.array_equal(bits(got), bits(total))


def test_reset_and_a_declared_pass_count_forget_the_group_in_progress(scenes):
    sc = scenes["spheres_a169"]
    with HipRenderer(sc, 200, 120, seed=SEED, exact=True) as r:
        want = r.render(6).radiance().copy()
        r.render(3).wait()  # leaves a group in progress behind
        r.reset()
        assert np.array_equal(bits(r.render(6).radiance()), bits(want))
        # continuing from a declared count inside a group: the buffer as it stands is the total, the group starts over
        r.set_pass_count(6)
        a = r.render(2).radiance().copy()
    with HipRenderer(sc, 200, 120, seed=SEED, exact=True) as r:
        r.render(6).wait()
        r.set_pass_count(6)
        b = r.render(1).render(1).radiance().copy()
    assert np.array_equal(bits(a), bits(b))


@pytest.mark.skipif(not os.path.exists(BIN), reason="kajo_render not built")
@pytest.mark.parametrize("numerics", ["--exact", "--fast"])
def test_kajo_render_batches_do_not_change_the_frame(tmp_path, numerics):
    """hip::Scheduler through the headless driver: 16 passes as batches of 4, 8, 16, 3 and as whatever its clock chooses."""
    frames = {}
    for name, extra in (("auto", ()), ("4", ("--batch", "4")), ("8", ("--batch", "8")), ("16", ("--batch", "16")), ("3", ("--batch", "3")),
                        ("3x3", ("--batch", "3", "--gpus", "3", "--same-device"))):
        raw = str(tmp_path / ("o_%s.raw" % name))
        cmd = [BIN, "-w", "1280", "-h", "720", "-r", "hip", "--passes", "16", "--raw", raw, "--json", numerics, *extra,
               os.path.join(ROOT, "kajo_amd", "data", "caustics.json")]  # (three lights: the kernel instances of any number of lights)
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr
        frames[name] = np.fromfile(raw, np.float32)
        os.remove(raw)
    for name, f in frames.items():
        assert np.array_equal(bits(f), bits(frames["16"])), (numerics, name)


@pytest.mark.parametrize("mode", ["fast", "exact"])
def test_four_wave_workgroups_with_a_parted_tail(scenes, mode):
    """A scene whose LDS copy exceeds 6 KB runs four-wave workgroups that share one copy (capi.cpp wavesPerBlock): 30 spheres / 2 lights,
    no grid. The parted tail's side-buffer slots, the slot base in LDS and the lanes' running totals are per WORKGROUP there: the same
    cuts, parts and orders must give one buffer; STRICT of the same scene is the oracle on a small frame."""
    from kajo_amd.scene import stress_scene
    sc = stress_scene(scenes["spheres_a169"], 30, 2, seed=11)
    kw = dict(exact=True) if mode == "exact" else {}
    want, tail = render_cuts(sc, 1280, 720, (16,), warm=1, **kw)
    assert tail > 0 and tail % 3 == 0  # parted: 3 more workgroups per parted block of a 16-pass launch
    for cuts in ((8, 8), (1, 2, 13), (12, 4)):
        got, _ = render_cuts(sc, 1280, 720, cuts, warm=1, **kw)
        assert np.array_equal(bits(got), bits(want)), (mode, cuts)
    for flags in (capi.KAJO_FLAG_NO_SPLIT, capi.KAJO_FLAG_NO_REORDER):
        got, tail = render_cuts(sc, 1280, 720, (16,), warm=1, flags=flags, **kw)
        assert tail == 0 and np.array_equal(bits(got), bits(want)), (mode, flags)


def test_four_wave_workgroups_strict_is_the_oracle(scenes):
    from kajo_amd.scene import stress_scene
    from oraclelib import OracleLib, available
    if not available("oracle"):
        pytest.skip("oracle not built")
    sc = stress_scene(scenes["spheres_a169"], 30, 2, seed=11)
    W, H, P = 160, 90, 3
    want = OracleLib("oracle").create(sc, 1).render(W, H, S=16, passes=P, seed=SEED, depth_limit=8)
    with HipRenderer(sc, W, H, spp=16, seed=SEED, strict=True) as r:
        got = r.render(P).radiance()
    same = (bits(got[..., :3]) == bits(want[..., :3])) | (np.isnan(got[..., :3]) & np.isnan(want[..., :3]))
    assert same.all(), int((~same).sum())
    with HipRenderer(sc, W, H, spp=16, seed=SEED, exact=True) as r:
        ex = r.render(P).radiance()
    from exact_tol import assert_exact_within_tolerance
    assert_exact_within_tolerance(ex, want, P, "30 spheres / 2 lights")


@pytest.mark.parametrize("mode", ["fast", "exact"])
def test_random_cuts_and_launch_sizes(scenes, mode):
    """Seeded fuzz: 24 passes cut at random into render calls, with random passesPerLaunch (so that a call is itself cut into launches that
    begin and end anywhere in a group), on a frame the SPLIT kernels render and on one the unsplit kernels render with a parted tail."""
    sc = scenes["caustics_a169"]  # three lights: the instances of any number of lights
    kw = dict(exact=True) if mode == "exact" else {}
    rng = np.random.default_rng(20261005 if mode == "exact" else 5)
    for W, H, cases in ((400, 240, 8), (1280, 720, 4)):
        with HipRenderer(sc, W, H, seed=SEED, passes_per_launch=24, **kw) as r:
            want = r.render(24).radiance()[..., :3].copy()
        for _ in range(cases):
            k = int(rng.integers(1, 7))
            cuts = np.sort(rng.choice(np.arange(1, 24), size=k - 1, replace=False)) if k > 1 else np.array([], int)
            sizes = np.diff(np.concatenate([[0], cuts, [24]])).tolist()
            ppl = int(rng.choice([0, 1, 3, 4, 5, 8, 16]))
            with HipRenderer(sc, W, H, seed=SEED, passes_per_launch=ppl, **kw) as r:
                if rng.random() < 0.5:  # a warmed handle: cost order and parts from the first launch on
                    r.render(2).wait()
                    r.reset()
                for s in sizes:
                    r.render(int(s))
                got = r.radiance()[..., :3].copy()
            assert np.array_equal(bits(got), bits(want)), (mode, W, H, sizes, ppl)
