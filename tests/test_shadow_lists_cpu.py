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
