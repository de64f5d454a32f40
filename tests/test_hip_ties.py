"""Exact ties and not-a-number samples, CONSTRUCTED (round 4's review: "no such tie occurs among these scenes" is an absence of
evidence). The reference's closest-hit walk accepts an object unless its distance is LARGER than the closest so far
(renderer/cpu/Raytracer.cpp:115), in scene order -- planes, then spheres (:131-132) -- so among bit-identical distances the LATER
object wins: a sphere over a plane, the higher-index one of two coincident spheres. Every walk of every numerics build must do
the same: the every-object walk of small scenes, the uniform-grid walk of large ones (whose cells deliver spheres in another order;
FAST gave ties to the sphere met first there in rounds 3-4 and packs (distance, ~index) into one 64-bit key since round 5), and
the visibility-list shadow query. The rays are axis-parallel with small-integer coordinates and the transforms are exact, so every
implementation computes the SAME distances and the tie is a tie in each of them."""
import numpy as np
import pytest

from kajo_amd import capi
from kajo_amd.renderer import HipRenderer
from kajo_amd.scene import Scene, material, plane_record, sphere_record, translate
from oraclelib import OracleLib, available

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not available("oracle"), reason="oracle not built")]

lin = lambda c: np.float32(c) ** np.float32(2.2)


def wall_at_x(x):
    """A plane (local y = 0, Raytracer.cpp:74-98) whose local y axis is the world's x axis, through (x, 0, 0): a rotation by a
    quarter turn with entries 0 / +-1 (exact inverse, determinant exactly 1), column-major."""
    M = np.array([[0, 1, 0, x], [-1, 0, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]], np.float32)
    return M.T.reshape(16).copy()


def tie_scene(base, fillers):
    """wall x = 2; sphere A centre (3, 0, 0) r = 1 (touches the wall's plane at (2, 0, 0)); spheres B and C COINCIDENT, centre
    (0, 4, 0) r = 1, different materials; `fillers` small spheres far from the test rays (>= 48 spheres enable the uniform grid)."""
    red, green, blue = (material(diffuse=[lin(.8), lin(.1), lin(.1)]), material(diffuse=[lin(.1), lin(.8), lin(.1)]),
                        material(diffuse=[lin(.1), lin(.1), lin(.8)]))
    recs = [sphere_record(translate(3, 0, 0), red, 1.0), sphere_record(translate(0, 4, 0), green, 1.0), sphere_record(translate(0, 4, 0), blue, 1.0)]
    rng = np.random.default_rng(11)
    for k in range(fillers):
        recs.append(sphere_record(translate(-6.0 - (k % 8), -3.0 - (k // 8) % 4, 2.0 + (k // 32)), material(diffuse=[lin(rng.uniform(.2, .8))] * 3), 0.25))
    recs.append(sphere_record(translate(-2, -2, -2), material(emission=[lin(16.0)] * 3), 0.25))
    planes = np.stack([plane_record(wall_at_x(2.0), material(diffuse=[lin(.5)] * 3))])
    return Scene(base.background, base.view, base.proj, np.stack(recs), planes, "ties%d" % fillers)


RAYS_O = np.array([[0, 0, 0], [0, 0, 0], [0, 8, 0]], np.float32)
RAYS_D = np.array([[1, 0, 0], [0, 1, 0], [0, -1, 0]], np.float32)
# ray 0: wall (object 1) and sphere A (object 2) both at t = 2 -> sphere A; rays 1, 2: spheres B (3) and C (4) coincide at t = 3 -> C
WANT_IDX = np.array([2, 4, 4], np.int32)


@pytest.mark.parametrize("fillers,flags,what", [(0, 0, "every-object walk"), (60, 0, "uniform grid"), (60, capi.KAJO_FLAG_NO_GRID, "every-object walk of the large scene")])
def test_bit_identical_distances_go_to_the_later_object(scenes, fillers, flags, what):
    sc = tie_scene(scenes["spheres_a169"], fillers)
    want = OracleLib("oracle").create(sc, 0).trace(RAYS_O, RAYS_D)
    assert np.array_equal(want["idx"], WANT_IDX), want["idx"]
    if available("ref_strict"):  # the compiled reference itself (oracle/_ref, where it travelled): the rule is ITS rule
        ref = OracleLib("ref_strict").create(sc).trace(RAYS_O, RAYS_D)
        assert np.array_equal(ref["idx"], WANT_IDX) and np.array_equal(ref["t"], want["t"])
    assert np.array_equal(want["t"], np.array([2, 3, 3], np.float32))
    for kw in ({"strict": True}, {}):  # (the EXACT build's walk is the STRICT build's)
        with HipRenderer(sc, 8, 8, flags=flags, **kw) as r:
            got = r.kat_trace(RAYS_O, RAYS_D)
        assert np.array_equal(got["idx"], WANT_IDX), (what, kw, got["idx"])
        assert np.array_equal(got["t"], want["t"]), (what, kw, got["t"])


def test_frames_of_the_tie_scene(scenes):
    """The same rule seen through whole paths: coincident spheres of different colours, a sphere touching a wall -- STRICT == oracle
    bit for bit, EXACT decides as the oracle does, FAST with the grid == FAST walking every sphere bit for bit."""
    from test_hip_parity import bits_equal
    base = scenes["spheres_a169"]
    for fillers in (0, 60):
        sc = tie_scene(base, fillers)
        W, H = 96, 54
        want = OracleLib("oracle").create(sc, 1).render(W, H, S=16, passes=2, seed=0o715517, depth_limit=8)
        with HipRenderer(sc, W, H, spp=16, strict=True) as r:
            got = r.render(2).radiance()
        assert ((got[..., :3].view(np.uint32) == want[..., :3].view(np.uint32)) | (np.isnan(got[..., :3]) & np.isnan(want[..., :3]))).all()
        with HipRenderer(sc, W, H, spp=16, exact=True) as r:
            ex = r.render(2).radiance()
        m = np.isfinite(want[..., :3]).all(-1)
        assert np.array_equal(np.isfinite(ex[..., :3]).all(-1), m)
        assert np.abs(ex[..., :3][m] - want[..., :3][m]).max() <= 1e-4 * max(1.0, np.abs(want[..., :3][m]).max())
        if fillers:
            with HipRenderer(sc, W, H, spp=16) as r:
                a = r.render(2).radiance()
            with HipRenderer(sc, W, H, spp=16, flags=capi.KAJO_FLAG_NO_GRID) as r:
                b = r.render(2).radiance()
            assert bits_equal(a, b)
