"""The launch tail (kajo_amd/csrc/capi.cpp partTheTail, integrator.inc.hip PARTS, aux_kernels.hip kajo_fold_parts).

From the second launch on -- the first measures the blocks -- the FAST and EXACT builds render the cheapest blocks of a large frame
as 2 or 4 workgroups of half / a quarter of the launch's passes each, so that the launch ends on short jobs (+2.3 % at 1920x1080,
profiles/r05_notes.txt). What that may change: a parted block's pixel total is  (total so far + passes of part 0) + passes of part 1
+ ...  instead of the passes' terms added one by one (Renderer.cpp:70-71) -- the same terms, each rendered from the same streams, in
another order of float additions: last places. What it must not change: anything else; pixels of blocks that are not parted keep
their bits; the result does not depend on which workgroup finishes first; STRICT -- the oracle's sum, term by term -- is never parted."""
import numpy as np
import pytest

from kajo_amd import capi
from kajo_amd.renderer import HipRenderer

pytestmark = pytest.mark.gpu

SEED = 0o715517
W, H = 1280, 720  # 14 400 pixel blocks: more than two rounds of the chip's 5 120 wave slots, the least the tail is parted for


def two_launches(sc, passes=8, **kw):
    with HipRenderer(sc, W, H, seed=SEED, passes_per_launch=passes, **kw) as r:
        r.render(passes).wait()
        first = r.counters()["tailGroups"]
        r.render(passes).wait()
        return r.radiance()[..., :3].copy(), first, r.counters()["tailGroups"]


@pytest.mark.parametrize("mode", ["fast", "exact"])
def test_parted_tail_is_the_unparted_sum_to_the_last_places(scenes, mode):
    sc = scenes["spheres_a169"]
    kw = dict(exact=True) if mode == "exact" else {}
    a, first, second = two_launches(sc, **kw)
    assert first == 0 and second > 0, (first, second)  # the first launch has no measured order yet
    b, f0, s0 = two_launches(sc, flags=capi.KAJO_FLAG_NO_SPLIT, **kw)
    assert f0 == 0 and s0 == 0
    nan_a, nan_b = ~np.isfinite(a).all(-1), ~np.isfinite(b).all(-1)
    assert np.array_equal(nan_a, nan_b)
    ok = ~nan_a
    same = (a == b).all(-1) | nan_a
    # 2 560 of the 14 400 blocks are parted (a quarter of the wave slots in four parts, a quarter in two): everything else keeps its bits
    assert same.mean() >= 0.80, same.mean()
    assert same.mean() < 1.0  # (if nothing differs the parts did not run)
    rel = np.abs(a - b)[ok] / np.maximum(np.abs(b[ok]), 1e-3)
    assert rel.max() <= 2e-6, rel.max()  # sums of 16 terms in another order: a few units in the last place
    # whole 8x8 blocks differ or do not
    blocks = same[: H // 8 * 8, : W // 8 * 8].reshape(H // 8, 8, W // 8, 8)
    parted = ~blocks.all(axis=(1, 3))
    assert 0 < parted.sum() <= 2560, parted.sum()
    # ... and the same bits whichever workgroup of a block finishes first: a second handle
    c, _, _ = two_launches(sc, **kw)
    assert np.array_equal(a, c, equal_nan=True)


def test_strict_is_never_parted(scenes):
    sc = scenes["spheres_a169"]
    a, first, second = two_launches(sc, strict=True)
    assert first == 0 and second == 0
    b, _, _ = two_launches(sc, strict=True, flags=capi.KAJO_FLAG_NO_SPLIT)
    assert np.array_equal(a, b, equal_nan=True)


def test_launches_whose_passes_do_not_divide_are_not_parted(scenes):
    """Parts are halves and quarters of the launch's passes, two passes at least: 6 passes are rendered whole."""
    sc = scenes["spheres_a169"]
    with HipRenderer(sc, W, H, seed=SEED, passes_per_launch=16, exact=True) as r:
        r.render(8).wait()
        r.render(6).wait()
        assert r.counters()["tailGroups"] == 0
        r.render(4).wait()  # quarters of one pass: no
        assert r.counters()["tailGroups"] == 0
        r.render(16).wait()
        assert r.counters()["tailGroups"] > 0
        got = r.radiance()[..., :3]
    with HipRenderer(sc, W, H, seed=SEED, passes_per_launch=16, exact=True, flags=capi.KAJO_FLAG_NO_SPLIT) as r:
        for p in (8, 6, 4, 16):
            r.render(p).wait()
        want = r.radiance()[..., :3]
    ok = np.isfinite(want).all(-1)
    assert np.array_equal(ok, np.isfinite(got).all(-1))
    assert (np.abs(got - want)[ok] / np.maximum(np.abs(want[ok]), 1e-3)).max() <= 2e-6


def test_large_scene_kernels_part_their_tail_too(scenes):
    """The 300-sphere scene (uniform grid, no lists) and the 1000-sphere one (grid + visibility lists) at 1280x720: four waves per
    SIMD there, 4 096 slots."""
    from kajo_amd.scene import stress_scene
    for sc in (stress_scene(scenes["spheres_a169"], 300, 6, seed=7), stress_scene(scenes["spheres_a169"], 1000, 16)):
        a, first, second = two_launches(sc, exact=True)
        assert first == 0 and second > 0
        b, _, _ = two_launches(sc, exact=True, flags=capi.KAJO_FLAG_NO_SPLIT)
        ok = np.isfinite(b).all(-1)
        assert np.array_equal(ok, np.isfinite(a).all(-1))
        assert (np.abs(a - b)[ok] / np.maximum(np.abs(b[ok]), 1e-3)).max() <= 2e-6
