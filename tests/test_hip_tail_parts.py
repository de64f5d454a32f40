"""The launch tail (kajo_amd/csrc/capi.cpp partTheTail, integrator.inc.hip GROUPS / PARTS, aux_kernels.hip kajo_fold_parts).

From the second launch on -- the first measures the blocks -- the FAST and EXACT builds render the cheapest blocks of a large frame of
a small scene as one workgroup per GROUP of four passes of the launch (a launch of 16 passes: four workgroups of four passes), so that
the launch ends on short jobs (+2 % at 1920x1080, profiles/r05_notes.txt). That must not change a bit of the frame: in these builds the
pixel's total takes the passes in groups of four by their absolute numbers -- each summed from zero in pass order, the group sums added
in group order -- whoever renders them: one wave all of them, several waves of a small frame dividing the passes (the SPLIT kernels), or
the workgroups of a parted block. So the frame does not depend on which blocks were parted, on the launch order, on how many GPUs shared
the frame, or on how the passes were cut into launches (tests/test_hip_pass_cuts.py). STRICT adds the passes' terms one by one, as the
oracle does (Renderer.cpp:70-71), and is never parted."""
import numpy as np
import pytest

from kajo_amd import capi
from kajo_amd.renderer import HipRenderer

pytestmark = pytest.mark.gpu

SEED = 0o715517
W, H = 1280, 720  # 14 400 pixel blocks: more than two rounds of the chip's 5 120 wave slots, the least the tail is parted for


def launches(sc, passes=(8, 8), w=W, h=H, **kw):
    """-> radiance after the launches, tailGroups after each"""
    groups = []
    with HipRenderer(sc, w, h, seed=SEED, passes_per_launch=max(passes), **kw) as r:
        for p in passes:
            r.render(p).wait()
            groups.append(r.counters()["tailGroups"])
        return r.radiance()[..., :3].copy(), groups


@pytest.mark.parametrize("mode", ["fast", "exact"])
def test_parted_tail_does_not_change_a_bit(scenes, mode):
    sc = scenes["spheres_a169"]
    kw = dict(exact=True) if mode == "exact" else {}
    a, g = launches(sc, (16, 16, 8), **kw)
    assert g[0] == 0 and g[1] > 0 and g[2] > 0, g  # the first launch has no measured order yet
    assert g[1] % 3 == 0 and 1000 <= g[1] // 3 <= 14400 // 2  # three more workgroups for every parted block
    assert g[2] == g[1] // 3  # (8 passes: two groups, one more workgroup per parted block)
    b, g0 = launches(sc, (16, 16, 8), flags=capi.KAJO_FLAG_NO_SPLIT, **kw)
    assert g0 == [0, 0, 0]
    assert np.array_equal(a, b, equal_nan=True)
    # ... nor does the order of the launch: workgroups in image order
    c, _ = launches(sc, (16, 16, 8), flags=capi.KAJO_FLAG_NO_REORDER, **kw)
    assert np.array_equal(a, c, equal_nan=True)


@pytest.mark.parametrize("mode", ["fast", "exact"])
def test_small_frames_divided_among_waves_form_the_same_sums(scenes, mode):
    """A 256 x 144 crop-sized frame runs the SPLIT kernels (several waves share a pixel block and divide the passes; wave 0 adds the
    terms): the same groups. Compared with the same frame rendered without the division."""
    sc = scenes["spheres_a169"]
    kw = dict(exact=True) if mode == "exact" else {}
    for passes in ((16,), (8, 16), (4, 6, 8)):
        a, _ = launches(sc, passes, w=256, h=144, **kw)
        b, _ = launches(sc, passes, w=256, h=144, flags=capi.KAJO_FLAG_NO_SPLIT, **kw)
        assert np.array_equal(a, b, equal_nan=True), passes


def test_groups_are_sums_of_the_same_terms(scenes):
    """What the groups change against adding pass by pass: the order of float additions, nothing else. EXACT's 16-pass frame against the
    float64 sum of its sixteen passes' own terms (each rendered alone into a zeroed buffer)."""
    sc = scenes["spheres_a169"]
    with HipRenderer(sc, 320, 180, seed=SEED, exact=True, flags=capi.KAJO_FLAG_NO_SPLIT) as r:
        a = r.render(16).radiance()[..., :3].copy()
        b = np.zeros(a.shape, np.float64)
        for p in range(16):
            r.reset()
            r.set_pass_count(p)
            b += r.render(1).radiance()[..., :3]
    ok = np.isfinite(b).all(-1)
    assert np.array_equal(ok, np.isfinite(a).all(-1))
    rel = np.abs(a - b)[ok] / np.maximum(np.abs(b[ok]), 1e-3)
    assert 0 < rel.max() <= 2e-6, rel.max()


def test_strict_is_never_parted(scenes):
    sc = scenes["spheres_a169"]
    a, g = launches(sc, (16, 16), strict=True)
    assert g == [0, 0]
    b, _ = launches(sc, (16, 16), strict=True, flags=capi.KAJO_FLAG_NO_SPLIT)
    assert np.array_equal(a, b, equal_nan=True)
    # pass by pass: the launches' boundaries do not matter to STRICT
    c, _ = launches(sc, (8, 6, 2, 16), strict=True)
    assert np.array_equal(a, c, equal_nan=True)


def test_only_launches_of_whole_groups_are_parted(scenes):
    """A launch is parted when it is two to eight whole groups of four passes (pass numbers decide where a group ends, so that every GPU of
    a frame forms the same sums); a launch that begins or ends inside a group, or is one group, is rendered whole."""
    sc = scenes["spheres_a169"]
    seq = (8, 6, 2, 16, 12, 4, 16, 6, 16, 2, 36)  # passes done before each: 0, 8, 14, 16, 32, 44, 48, 64, 70, 86, 88
    with HipRenderer(sc, W, H, seed=SEED, passes_per_launch=64, exact=True) as r:
        g = []
        for p in seq:
            r.render(p).wait()
            g.append(r.counters()["tailGroups"])
        a = r.radiance()[..., :3].copy()
    assert [x > 0 for x in g] == [False, False, False, True, True, False, True, False, False, False, False], g
    assert g[3] % 3 == 0 and g[4] == g[3] // 3 * 2  # one more workgroup per parted block and group beyond the first
    with HipRenderer(sc, W, H, seed=SEED, passes_per_launch=64, exact=True, flags=capi.KAJO_FLAG_NO_SPLIT) as r:
        for p in seq:
            r.render(p)
        b = r.radiance()[..., :3].copy()
    assert np.array_equal(a, b, equal_nan=True)
    with HipRenderer(sc, W, H, seed=SEED, passes_per_launch=64, exact=True) as r:
        c = r.render(sum(seq)).radiance()[..., :3].copy()  # 124 passes: 64 (16 groups: whole) + 60
    assert np.array_equal(a, c, equal_nan=True)


def test_large_scenes_are_not_parted(scenes):
    """The large-scene kernels (uniform grid; cold records in global memory) add pass by pass in every build."""
    from kajo_amd.scene import stress_scene
    sc = stress_scene(scenes["spheres_a169"], 300, 6, seed=7)
    a, g = launches(sc, (8, 8), exact=True)
    assert g == [0, 0]
    b, _ = launches(sc, (4, 4, 8), exact=True)
    assert np.array_equal(a, b, equal_nan=True)
