"""Host-side checks of the per-light visibility lists of large scenes (kajo_amd/csrc/stage.cpp buildShadowLists,
device_scene.h DShadowLists): the candidate set a shadow query tests must contain every sphere that the brute-force
closest-hit walk (the oracle's restatement of Raytracer.cpp:100-138) finds in front of the light -- for shadow rays formed the
way the integrator forms them (Shader.cpp:59-66: origin = vertex + direction * epsilon, direction toward a point of the light).
No GPU: the lists are staged on the host; the kernels' use of them is covered by the STRICT = oracle frame tests."""
import numpy as np
import pytest

from kajo_amd.renderer import stage_shadow_lists
from kajo_amd.scene import stress_scene
from oraclelib import OracleLib, available

pytestmark = pytest.mark.skipif(not available("oracle"), reason="oracle not built")


def bin_of(u, n):
    """integrator.inc.hip lightReached: cube-map bin of u = O - C, in binary32 like the kernel."""
    u = u.astype(np.float32)
    a = np.abs(u)
    m = np.where((a[:, 0] >= a[:, 1]) & (a[:, 0] >= a[:, 2]), 0, np.where(a[:, 1] >= a[:, 2], 1, 2))
    r = np.arange(len(u))
    um, ua, ub = u[r, m], u[r, (m + 1) % 3], u[r, (m + 2) % 3]
    im = np.float32(1) / np.abs(um)
    half = np.float32(0.5 * n)
    ia = np.clip(np.floor((ua * im + np.float32(1)) * half).astype(np.int64), 0, n - 1)
    ib = np.clip(np.floor((ub * im + np.float32(1)) * half).astype(np.int64), 0, n - 1)
    face = 2 * m + (um < 0)
    return (face * n + ib) * n + ia


@pytest.mark.parametrize("n_spheres,n_lights,seed", [(1000, 16, 1234), (300, 6, 7), (60, 2, 3)])
def test_every_blocker_is_a_candidate(scenes, n_spheres, n_lights, seed):
    sc = stress_scene(scenes["spheres_a169"], n_spheres, n_lights, seed=seed)
    L = stage_shadow_lists(sc)
    assert L is not None and len(L["lights"]) == n_lights
    n, start, key, index = L["n"], L["start"], L["key"], L["index"]
    assert np.all(np.diff(start.astype(np.int64)) >= 0) and start[-1] == len(key)
    for b in range(0, len(start) - 1, 97):  # keys ascend within a bin
        assert np.all(np.diff(key[start[b]:start[b + 1]]) >= 0)
    h = OracleLib("oracle").create(sc, 1)
    rng = np.random.default_rng(seed)
    # vertices: closest hits of random rays from random points of the room
    m = 6000
    org = np.stack([rng.uniform(-3, 9, m), rng.uniform(-1.8, 0.9, m), rng.uniform(-1.5, 4.5, m)], 1).astype(np.float32)
    d = rng.normal(size=(m, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    hit = h.trace(org, d)
    ok = hit["idx"] > 0
    P = hit["position"][ok]
    np_ = sc.n_planes
    centre = sc.spheres[:, 12:15]
    radius = sc.spheres[:, 38]
    blocked = tested = 0
    for k, sl in enumerate(L["lights"]):
        C, r = centre[sl], radius[sl]
        q = rng.normal(size=P.shape).astype(np.float32)
        q *= (r * rng.random((len(P), 1)) ** (1 / 3) / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
        l = (C + q - P).astype(np.float32)
        l /= np.linalg.norm(l, axis=1, keepdims=True).astype(np.float32)
        O = (P + l * np.float32(1e-3)).astype(np.float32)
        got = h.trace(O, l)["idx"]
        u = (O - C).astype(np.float32)
        bins = k * 6 * n * n + bin_of(u, n)
        reach = np.maximum(np.sqrt((u.astype(np.float32) ** 2).sum(1)), r) * np.float32(1.000001)
        for j in np.nonzero((got > np_) & (got != np_ + 1 + sl))[0]:  # a sphere other than the light is the closest hit
            lo, hi = start[bins[j]], start[bins[j] + 1]
            cand = index[lo:hi][key[lo:hi] <= reach[j]]
            assert got[j] - 1 - np_ in cand, (k, j, got[j], cand)
            blocked += 1
        lo, hi = start[bins], start[bins + 1]
        tested += sum(int((key[a:b] <= rc).sum()) for a, b, rc in zip(lo, hi, reach))
    assert blocked > 50
    print("%d spheres / %d lights: %d blocked rays checked, %.2f candidate spheres per query (of %d)" % (
        n_spheres, n_lights, blocked, tested / (len(P) * n_lights), n_spheres))


def test_small_scenes_get_no_lists(scenes):
    assert stage_shadow_lists(scenes["spheres_a169"]) is None  # 5 spheres: no grid, every object is walked


def adversarial_scene(base, seed, n=90, n_lights=5):
    """Geometry the lists' margins have to survive: overlapping and nested spheres, big lights, lights that touch or contain
    other spheres, spheres cut by the room's planes, a light hugging a wall, tiny spheres; all pure translations (balls)."""
    from kajo_amd.scene import Scene, material, sphere_record, translate
    rng = np.random.default_rng(seed)
    lin = lambda c: np.float32(c) ** np.float32(2.2)
    recs = []
    for k in range(n):
        c = np.array([rng.uniform(-4, 8), rng.uniform(-1.9, 0.95), rng.uniform(-1.8, 5)])
        r = float(rng.choice([0.01, 0.05, 0.2, 0.5, 0.9], p=[.1, .3, .3, .2, .1]))
        if k % 7 == 3 and recs:  # nested in / overlapping the previous one
            c = recs[-1][12:15] + rng.normal(size=3) * 0.1
        col = [lin(.2 + .6 * rng.random()) for _ in range(3)]
        m = material(diffuse=col) if k % 2 == 0 else material(specular=col, exponent=float(rng.choice([0, 10, 100])))
        recs.append(sphere_record(translate(*c), m, r))
    for k in range(n_lights):
        c = [rng.uniform(-3, 7), rng.uniform(-1.9, 0.5), rng.uniform(-1, 4)]
        r = float(rng.choice([0.05, 0.3, 0.8]))
        if k == 0:
            c, r = [2.0, -1.95, 1.0], 0.3  # cut by the ceiling plane (y = -2)
        if k == 1 and recs:
            c, r = list(recs[0][12:15] + np.array([recs[0][38] + 0.1, 0, 0])), 0.1  # touching sphere 0
        recs.append(sphere_record(translate(*c), material(emission=[lin(6.0)] * 3), r))
    return Scene(base.background, base.view, base.proj, np.stack(recs), base.planes, "adversarial%d" % seed)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_blockers_are_candidates_on_adversarial_geometry(scenes, seed):
    sc = adversarial_scene(scenes["spheres_a169"], seed)
    L = stage_shadow_lists(sc)
    assert L is not None
    n, start, key, index = L["n"], L["start"], L["key"], L["index"]
    h = OracleLib("oracle").create(sc, 1)
    rng = np.random.default_rng(seed)
    m = 5000
    org = np.stack([rng.uniform(-3, 9, m), rng.uniform(-1.8, 0.9, m), rng.uniform(-1.5, 4.5, m)], 1).astype(np.float32)
    d = rng.normal(size=(m, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    hit = h.trace(org, d)
    P = hit["position"][hit["idx"] > 0]
    np_, centre, radius = sc.n_planes, sc.spheres[:, 12:15], sc.spheres[:, 38]
    checked = 0
    for k, sl in enumerate(L["lights"]):
        C, r = centre[sl], radius[sl]
        q = rng.normal(size=P.shape).astype(np.float32)
        q *= (r * rng.random((len(P), 1)) ** (1 / 3) / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
        l = (C + q - P).astype(np.float32)
        l /= np.linalg.norm(l, axis=1, keepdims=True).astype(np.float32)
        O = (P + l * np.float32(1e-3)).astype(np.float32)
        got = h.trace(O, l)["idx"]
        u = (O - C).astype(np.float32)
        bins = k * 6 * n * n + bin_of(u, n)
        reach = np.maximum(np.sqrt((u ** 2).sum(1)), r) * np.float32(1.000001)
        # also: the light itself must be hit along such a ray whenever the walk says it is the closest hit (nothing to check
        # in the lists), and any OTHER sphere that is the closest hit must be listed
        for j in np.nonzero((got > np_) & (got != np_ + 1 + sl))[0]:
            lo, hi = start[bins[j]], start[bins[j] + 1]
            assert got[j] - 1 - np_ in index[lo:hi][key[lo:hi] <= reach[j]], (seed, k, j)
            checked += 1
    assert checked > 500


def test_room_detection(scenes):
    """stage.cpp findRoom: the culling structures' margins are sized from where a ray can START (what binary32 rounding lets the
    reference's sphere test report as a hit grows with |O - c|^2, Raytracer.cpp:26-30). A bounded convex region of opaque planes
    around the camera bounds it; anything else does not: then no visibility lists, and a grid that hands far rays to the
    every-sphere loop (round-4 advisor finding: the margins were sized from the spheres' own extent, and a vertex 100 units out on
    an open floor made the lists miss a quarter of the blockers)."""
    from kajo_amd.renderer import stage_info
    from kajo_amd.scene import Scene
    base = scenes["spheres_a169"]
    big = stress_scene(base, 300, 6, seed=7)
    info = stage_info(big)
    assert info["closed_room"] and info["grid"] and info["shadow_lists"] and info["grid_reach"] == 0.0
    # spheres.json's room (data/spheres.json:42-79): x in [-8, 10], y in [-2, 1], z in [-2, 6]
    assert np.allclose(info["room"], [-8, -2, -2, 10, 1, 6], atol=1e-3)
    # five of the six walls: open to one side
    for drop in range(6):
        sc = Scene(big.background, big.view, big.proj, big.spheres, np.delete(big.planes, drop, 0), "open%d" % drop)
        info = stage_info(sc)
        assert not info["closed_room"] and not info["shadow_lists"] and info["grid"] and info["grid_reach"] > 10.0, drop
        assert stage_shadow_lists(sc) is None
    # a floor only
    sc = Scene(big.background, big.view, big.proj, big.spheres, big.planes[:1], "floor")
    assert not stage_info(sc)["closed_room"] and stage_shadow_lists(sc) is None
    # a wall of glass: paths go on behind it
    planes = big.planes.copy()
    planes[3, 16 + 16:16 + 20] = (0.5, 0.5, 0.5, 1.0)  # Material.transparency (scene/Scene.h:11-23: ambient, diffuse, specular, emission, transparency)
    sc = Scene(big.background, big.view, big.proj, big.spheres, planes, "glasswall")
    assert stage_info(sc)["grid"] and not stage_info(sc)["closed_room"] and stage_shadow_lists(sc) is None
    # the room turned as a whole (no axis-parallel wall): still a room, its box the turned room's
    from scenes_extra import rotation
    R = rotation(np.random.default_rng(4)).astype(np.float64)
    planes = big.planes.copy()
    spheres = big.spheres.copy()
    for rec in (planes, spheres):
        for k in range(len(rec)):
            M = rec[k, :16].reshape(4, 4).T.astype(np.float64)  # column-major image -> matrix
            rec[k, :16] = (R @ M).astype(np.float32).T.reshape(16)
    view = (big.view.reshape(4, 4).T.astype(np.float64) @ np.linalg.inv(R)).astype(np.float32).T.reshape(16)
    sc = Scene(big.background, view, big.proj, spheres, planes, "turned")
    info = stage_info(sc)
    assert info["closed_room"]
    corners = np.array([[x, y, z, 1] for x in (-8, 10) for y in (-2, 1) for z in (-2, 6)], np.float64) @ R.T
    assert np.allclose(info["room"][:3], corners[:, :3].min(0), atol=1e-2) and np.allclose(info["room"][3:], corners[:, :3].max(0), atol=1e-2)
