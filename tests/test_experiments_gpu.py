"""Experiments kept for their evidence, built apart from the product (`make -C kajo_amd/csrc experiments` ->
kajo_amd/libkajo_hip_r02.so: round 2's kernels, with the cooperative-traversal `_coop` variants). The shipped
libkajo_hip.so contains none of them; these tests load the experiment library in a child process (KAJO_HIP_LIB)."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "kajo_amd", "libkajo_hip_r02.so")
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not os.path.exists(LIB), reason="experiment library not built (make -C kajo_amd/csrc experiments)")]

COOP = textwrap.dedent("""
    import sys
    sys.path[:0] = [%r, %r]
    import numpy as np
    from kajo_amd import capi
    from kajo_amd.renderer import HipRenderer
    from kajo_amd.scene import Scene
    from oraclelib import OracleLib
    from test_hip_parity import stress_scene, SEED, bits_equal
    assert capi.LIB_PATH.endswith("libkajo_hip_r02.so")
    z = np.load(%r)
    sc = stress_scene(Scene.from_npz(z, "spheres_a169/", "spheres_a169"), 300, 6, seed=7)
    W, H = 128, 64  # 2 x 4 tiles of 64 x 16 = 128 waves: sixteen 8-wave workgroups
    want = OracleLib("oracle").create(sc, math=1).render(W, H, S=4, passes=3, seed=SEED, depth_limit=8)
    with HipRenderer(sc, W, H, spp=4, seed=SEED, strict=True, flags=capi.KAJO_FLAG_COOP, passes_per_launch=2) as r:
        got = r.render(3).radiance()
    same = (got[..., :3].view(np.uint32) == want[..., :3].view(np.uint32)) | (np.isnan(got[..., :3]) & np.isnan(want[..., :3]))
    assert same.all()
    with HipRenderer(sc, W, H, spp=16, seed=SEED, flags=capi.KAJO_FLAG_COOP) as r:
        a = r.render(2).radiance()
    with HipRenderer(sc, W, H, spp=16, seed=SEED) as r:
        b = r.render(2).radiance()
    assert bits_equal(a, b)
    print("coop ok")
""") % (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden", "scenes.npz"))


def test_cooperative_sorted_traversal_is_result_identical():
    """KAJO_FLAG_COOP (round 2): the 8 waves of a workgroup pool their rays in LDS every trip, counting-sort
    them by ray kind and octant into a compact queue, and every lane walks the ray at its queue position. Which lane walks a ray
    does not change its arithmetic: STRICT stays the oracle bit for bit, FAST stays FAST. Measured slower; not shipped."""
    env = dict(os.environ, KAJO_HIP_LIB=LIB)
    p = subprocess.run([sys.executable, "-c", COOP], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "coop ok" in p.stdout, p.stdout + p.stderr
