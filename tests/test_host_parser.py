"""The host-side scene loader (kajo_amd/host/scene/SceneLoader.cpp) against the
reference's own parser: tests/golden/scenes.npz holds what scene::Parser::load produced for the
same files (both the -O2 and the fast-math build of the reference)."""
import ctypes as C
import os

import numpy as np
import pytest

from kajo_amd.scene import Scene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "kajo_amd", "libkajo_host.so")
pytestmark = pytest.mark.skipif(not os.path.exists(LIB), reason="libkajo_host.so not built")


def parse(text, aspect, name="parsed"):
    L = C.CDLL(LIB)
    ns, npl = C.c_int(), C.c_int()
    rc = L.kajo_host_parse(text.encode() if text is not None else None, C.c_float(aspect), C.byref(ns), C.byref(npl))
    if rc != 0:
        return None
    bg, view, proj = np.zeros(4, np.float32), np.zeros(16, np.float32), np.zeros(16, np.float32)
    sph, pl = np.zeros((ns.value, 39), np.float32), np.zeros((npl.value, 38), np.float32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    L.kajo_host_export(p(bg), p(view), p(proj), p(sph), p(pl))
    return Scene(bg, view, proj, sph, pl, name)


def compare(got: Scene, z, key):
    for tag, tol in (("strict_", 0.0), ("", 2e-6)):
        want = Scene.from_npz(z, key + "/" + tag)
        assert got.spheres.shape == want.spheres.shape and got.planes.shape == want.planes.shape
        for a, b, what in ((got.background, want.background, "background"), (got.view, want.view, "view"),
                           (got.proj, want.proj, "proj"), (got.spheres, want.spheres, "spheres"),
                           (got.planes, want.planes, "planes")):
            if tol == 0.0:
                assert np.array_equal(a, b), (key, what, np.abs(a - b).max())
            else:
                assert np.abs(a - b).max() <= tol * max(1.0, float(np.abs(b).max())), (key, what)


@pytest.mark.parametrize("key,rel,aspect", [("caustics_a169", "kajo_amd/data/caustics.json", 1920.0 / 1080.0),
                                             ("dialect_a1", "kajo_amd/data/dialect.json", 1.0)])
def test_own_scene_files_match_reference_parser(golden, key, rel, aspect):
    got = parse(open(os.path.join(ROOT, rel)).read(), aspect)
    assert got is not None
    compare(got, golden.scenes, key)


@pytest.mark.parametrize("key,rel,aspect", [("spheres_a1", "data/spheres.json", 1.0), ("spheres_a169", "data/spheres.json", 1920.0 / 1080.0),
                                             ("spheres_a43", "data/spheres.json", 640.0 / 480.0), ("test_a1", "data/test.json", 1.0)])
def test_reference_scene_files(golden, key, rel, aspect):
    path = os.path.join("/root/reference", rel)
    if not os.path.exists(path):
        pytest.skip("reference tree not present (GPU box)")
    got = parse(open(path).read(), aspect)
    assert got is not None
    compare(got, golden.scenes, key)


def test_rejects_malformed_input():
    assert parse("[1, 2, 3]", 1.0) is None      # not an object (Parser.cpp:220-221)
    assert parse("{\"objects\": [", 1.0) is None
    assert parse("", 1.0) is None
    empty = parse("{}", 1.0)
    assert empty is not None and empty.n_spheres == 0 and empty.n_planes == 0


def test_builtin_test_scene():
    """renderer/Main.cpp:13-95: 5 spheres (one emissive, radius .3, emission 8), 6 planes, 4:3 camera."""
    s = parse(None, 4.0 / 3.0)
    assert s.n_spheres == 5 and s.n_planes == 6 and s.n_lights == 1
    assert np.allclose(s.spheres[4, 16 + 12:16 + 16], [8, 8, 8, 0]) and np.isclose(s.spheres[4, 38], .3)
    assert np.allclose(s.spheres[0, 16 + 16:16 + 20], .9) and np.isclose(s.spheres[0, 37], 1.5)
    assert np.isclose(s.spheres[1, 36], 20)
    assert np.allclose(s.spheres[3, 12:15], [7, 0, 1.5])
    assert np.allclose(s.planes[0, 16 + 4:16 + 8], [.4, .4, .4, 1])
    assert np.isclose(s.proj[0], s.proj[5] / (4.0 / 3.0), rtol=1e-6)
