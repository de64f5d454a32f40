"""The launch tail (kajo_amd/csrc/capi.cpp partTheTail, integrator.inc.hip GROUPS / PARTS, aux_kernels.hip kajo_fold_parts).

From the second launch on -- the first measures the blocks -- the FAST and EXACT builds render the cheapest blocks of a large frame of
a small scene as FOUR workgroups of a quarter of the launch's passes each, so that the launch ends on short jobs (+2 % at 1920x1080,
profiles/r05_notes.txt). That must not change a bit of the frame: in these builds the pixel's total takes the passes of a launch of
8, 16, 32 ... passes in four groups -- each summed from zero in pass order, the group sums added in group order -- whoever renders them:
one wave all of them, several waves of a small frame dividing the passes (the SPLIT kernels), or the four workgroups of a parted block.
So the frame does not depend on which blocks were parted, on the launch order, or on how many GPUs shared the frame. STRICT adds the
passes' terms one by one, as the oracle does (Renderer.cpp:70-71), and is never parted."""
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
    """What the groups may change against adding pass by pass: the order of float additions. The EXACT frame of 16 passes in one launch
    (four groups) against the same 16 passes in launches of 2 (each launch one group: pass by pass up to a zero added first)."""
    sc = scenes["spheres_a169"]
    a, _ = launches(sc, (16,), w=320, h=180, exact=True, flags=capi.KAJO_FLAG_NO_SPLIT)
    b, _ = launches(sc, (2,) * 8, w=320, h=180, exact=True, flags=capi.KAJO_FLAG_NO_SPLIT)
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


def test_only_launches_of_8_16_32_passes_on_a_group_boundary_are_parted(scenes):
    """Groups are quarters of a launch of 8, 16, 32 ... passes that starts where a group of its size would (pass numbers decide where a
    group ends, so that every GPU of a frame forms the same sums); any other launch is one group, rendered whole."""
    sc = scenes["spheres_a169"]
    seq = (8, 6, 2, 16, 12, 4, 16, 6, 16)  # passes done before each: 0, 8, 14, 16, 32, 44, 48, 64, 70
    a, g = launches(sc, seq, exact=True)
    assert [x > 0 for x in g] == [False, False, False, True, False, False, True, False, False], g
    b, _ = launches(sc, seq, exact=True, flags=capi.KAJO_FLAG_NO_SPLIT)
    assert np.array_equal(a, b, equal_nan=True)


def test_large_scenes_are_not_parted(scenes):
    """The large-scene kernels (uniform grid; cold records in global memory) add pass by pass in every build."""
    from kajo_amd.scene import stress_scene
    sc = stress_scene(scenes["spheres_a169"], 300, 6, seed=7)
    a, g = launches(sc, (8, 8), exact=True)
    assert g == [0, 0]
    b, _ = launches(sc, (4, 4, 8), exact=True)
    assert np.array_equal(a, b, equal_nan=True)
