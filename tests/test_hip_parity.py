"""GPU parity tests (run on an MI355X: pytest -m gpu). Everything goes through the C ABI
(libkajo_hip.so); the oracle (oracle/libkajo_oracle.so) is the checker.

  STRICT kernels vs oracle(strict math): bit for bit -- every decision of every path.
  FAST kernels vs oracle(libm) and vs the golden frames of the compiled reference: the
  tolerance SURVEY.md section 8c states (median |d| <= 1e-5, p99 <= 2e-3, clamped RMSE <= 1e-3 on
  the radiance estimate), i.e. the floor the reference has between two builds of itself.
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

from kajo_amd.renderer import HipRenderer
from kajo_amd.scene import caustics_scene, stress_scene
from oraclelib import OracleLib, available

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not available("oracle"), reason="oracle not built")]

SEED = 0o715517


@pytest.fixture(scope="module")
def O():
    return OracleLib("oracle")


def frame_stats(a, b):
    m = np.isfinite(a) & np.isfinite(b)
    d = np.abs(a - b)[m]
    ca, cb = np.clip(a, 0, 1), np.clip(b, 0, 1)
    return dict(median=float(np.median(d)), p99=float(np.percentile(d, 99)), max=float(d.max()),
                clamped_rmse=float(np.sqrt(np.mean(((ca - cb) ** 2)[m]))), nonfinite=int((~m).sum()),
                identical=float(np.mean(d == 0)))


def bits_equal(a, b):
    return np.array_equal(np.ascontiguousarray(a, np.float32).view(np.uint32), np.ascontiguousarray(b, np.float32).view(np.uint32))


STRICT_CASES = [
    ("spheres_a1", 256, 256, 16, 1, 1),    # BASELINE configs[0] (C1) exactly: 256x256, 16 spp, 1 bounce
    # scene key, W, H, S, passes, depth
    ("spheres_a1", 64, 64, 16, 1, 1),      # BASELINE configs[0] shape (C1) at 64x64
    ("spheres_a1", 64, 64, 32, 2, 8),
    ("spheres_a169", 100, 37, 32, 1, 8),   # ragged: not a multiple of the tile or of 8
    ("test_a1", 48, 48, 32, 2, 8),
    ("spheres_a1", 8, 8, 1, 3, 8),         # one sample per pixel, three passes
    ("spheres_a1", 33, 1, 4, 1, 0),        # one row, depth limit 0
]


@pytest.mark.parametrize("key,W,H,S,passes,depth", STRICT_CASES)
def test_strict_bit_exact(O, scenes, key, W, H, S, passes, depth):
    sc = scenes[key]
    want = O.create(sc, math=1).render(W, H, S=S, passes=passes, seed=SEED, depth_limit=depth)
    with HipRenderer(sc, W, H, spp=S, depth_limit=depth, seed=SEED, strict=True) as r:
        got = r.render(passes).radiance()
    same = got[..., :3].view(np.uint32) == want[..., :3].view(np.uint32)
    nan_both = np.isnan(got[..., :3]) & np.isnan(want[..., :3])
    assert (same | nan_both).all(), "%d of %d channels differ; stats %s" % (
        (~(same | nan_both)).sum(), same.size, frame_stats(got[..., :3], want[..., :3]))


def test_strict_bit_exact_synthetic_scenes(O, scenes):
    base = scenes["spheres_a169"]
    for sc, W, H in ((caustics_scene(base), 64, 36), (stress_scene(base, 200, 8), 32, 18)):
        want = O.create(sc, math=1).render(W, H, S=16, passes=1, seed=SEED, depth_limit=8)
        with HipRenderer(sc, W, H, spp=16, seed=SEED, strict=True) as r:
            got = r.render(1).radiance()
        same = got[..., :3].view(np.uint32) == want[..., :3].view(np.uint32)
        nan_both = np.isnan(got[..., :3]) & np.isnan(want[..., :3])
        assert (same | nan_both).all(), (sc.name, frame_stats(got[..., :3], want[..., :3]))


def test_builtin_test_scene_strict(O):
    """renderer/Main.cpp:13-95 (the scene the reference builds without a scene file), as produced by the
    host-side loader: emission alpha 0, transparency (.9,.9,.9,.9), Phong exponent without specular colour."""
    import os
    from test_host_parser import LIB, parse
    if not os.path.exists(LIB):
        pytest.skip("libkajo_host.so not built")
    sc = parse(None, 4.0 / 3.0, "buildTestScene")
    W, H = 64, 48
    want = O.create(sc, math=1).render(W, H, S=16, passes=2, seed=SEED, depth_limit=8)
    with HipRenderer(sc, W, H, spp=16, seed=SEED, strict=True) as r:
        got = r.render(2).radiance()
    same = got[..., :3].view(np.uint32) == want[..., :3].view(np.uint32)
    nan_both = np.isnan(got[..., :3]) & np.isnan(want[..., :3])
    assert (same | nan_both).all(), frame_stats(got[..., :3], want[..., :3])
    with HipRenderer(sc, W, H, spp=16, seed=SEED) as r:
        fast = r.render(2).radiance()
    s = frame_stats(fast[..., :3] / 2, O.create(sc, math=0).render(W, H, S=16, passes=2, seed=SEED)[..., :3] / 2)
    assert s["median"] <= 1e-5 and s["p99"] <= 2e-3 and s["clamped_rmse"] <= 1e-3, s


def test_grid_walk_equals_brute_force(O, scenes):
    """Large scenes reach their spheres through a uniform grid (3D-DDA); the closest hit and the tie rule
    must be those of the walk over every sphere: STRICT grid == oracle (which tests every sphere) bit for
    bit. FAST grid == FAST brute force bit for bit: the same per-sphere arithmetic, and since round 5 the same tie rule -- the grid walk
    compares (distance pattern, ~index) as one 64-bit key, so that of two objects at a bit-identical distance the later one wins whatever
    order the cells deliver them in (tests/test_hip_ties.py constructs such ties)."""
    from kajo_amd import capi
    base = scenes["spheres_a169"]
    for sc, W, H in ((stress_scene(base, 300, 6, seed=7), 96, 54), (stress_scene(base, 1000, 16), 64, 36)):
        want = O.create(sc, math=1).render(W, H, S=4, passes=1, seed=SEED, depth_limit=8)
        with HipRenderer(sc, W, H, spp=4, seed=SEED, strict=True) as r:
            got = r.render(1).radiance()
        same = got[..., :3].view(np.uint32) == want[..., :3].view(np.uint32)
        nan_both = np.isnan(got[..., :3]) & np.isnan(want[..., :3])
        assert (same | nan_both).all(), (sc.name, frame_stats(got[..., :3], want[..., :3]))
        with HipRenderer(sc, W, H, spp=16, seed=SEED, counters=True) as g:
            a = g.render(1).radiance()
            cg = g.counters()
        b = HipRenderer(sc, W, H, spp=16, seed=SEED)
        b.params.flags |= 4  # KAJO_FLAG_NO_GRID only takes effect at create: build a second renderer
        b.close()
        import ctypes as C
        p = capi.KajoParams()
        capi.lib().kajo_hip_default_params(C.byref(p))
        p.samplesPerPass, p.seed, p.flags = 16, SEED, 4
        h = C.c_void_p()
        pod = sc.pod()
        capi.check(capi.lib().kajo_hip_create(C.byref(pod), W, H, C.byref(p), C.byref(h)))
        capi.check(capi.lib().kajo_hip_render(h, 1))
        brute = np.empty((H, W, 4), np.float32)
        capi.check(capi.lib().kajo_hip_read_radiance(h, brute.ctypes.data_as(C.c_void_p)))
        capi.lib().kajo_hip_destroy(h)
        assert bits_equal(a, brute), sc.name


def test_shadow_lists_equal_the_grid_walk(O, scenes):
    """Large scenes whose spheres are all balls answer a light sample's shadow query inside the light loop from the light's
    visibility lists (integrator.inc.hip lightReached) instead of giving the shadow ray a trip of its own through the grid
    (KAJO_FLAG_NO_SHADOW_LISTS: round 3's schedule). Same draws, same per-object arithmetic, same acceptance rule: STRICT is the
    oracle bit for bit either way, FAST the same buffer either way; the walk counts show where the shadow rays went."""
    from kajo_amd import capi
    base = scenes["spheres_a169"]
    for sc, W, H, S, passes in ((stress_scene(base, 300, 6, seed=7), 96, 54, 9, 2), (stress_scene(base, 1000, 16), 80, 45, 4, 3)):
        want = O.create(sc, math=1).render(W, H, S=S, passes=passes, seed=SEED, depth_limit=8)
        got = {}
        for strict in (True, False):
            for flags in (0, capi.KAJO_FLAG_NO_SHADOW_LISTS):
                with HipRenderer(sc, W, H, spp=S, seed=SEED, strict=strict, counters=True, flags=flags, passes_per_launch=2) as r:
                    got[strict, flags] = (r.render(passes).radiance(), r.counters())
        for flags in (0, capi.KAJO_FLAG_NO_SHADOW_LISTS):
            g = got[True, flags][0][..., :3]
            assert ((g.view(np.uint32) == want[..., :3].view(np.uint32)) | (np.isnan(g) & np.isnan(want[..., :3]))).all(), (sc.name, flags)
        assert bits_equal(got[False, 0][0], got[False, capi.KAJO_FLAG_NO_SHADOW_LISTS][0]), sc.name
        for strict in (True, False):
            a, b = got[strict, 0][1], got[strict, capi.KAJO_FLAG_NO_SHADOW_LISTS][1]
            assert b["shadowQueries"] == 0 and a["shadowQueries"] > 0
            assert a["traversals"] + a["shadowQueries"] == b["traversals"], (sc.name, strict, a, b)  # every shadow ray is one or the other
            assert a["vertices"] == b["vertices"]


@pytest.mark.parametrize("numerics", ["fast", "strict", "exact"])
def test_one_light_kernel_equals_the_general_one(scenes, numerics):
    """Small scenes with exactly one light run a kernel instance that samples the extension ray in the same visit of the
    light / BSDF blocks as the light (integrator.inc.hip PRESAMPLE) -- one visit per vertex instead of two (FAST), and the shadow
    ray in a trip of its own instead of a walk inside the light loop (STRICT). Same draws in the same order, same arithmetic: the
    buffer is the general instance's (KAJO_FLAG_NO_ONE_LIGHT) bit for bit, on every one-light scene, whole launches and split ones,
    every depth limit; vertices counted the same, and every shadow ray counted as a walk of the one kind or the other."""
    from kajo_amd import capi
    for key, W, H, S, passes, ppl, depth in (("spheres_a169", 320, 180, 32, 5, 2, 8), ("spheres_a1", 128, 128, 16, 1, 1, 1), ("test_a1", 160, 160, 9, 3, 0, 8),
                                             ("dialect_a1", 97, 61, 4, 4, 3, 3), ("spheres_a43", 200, 150, 32, 2, 0, 2)):
        sc = scenes[key]
        assert sc.n_lights == 1, key
        got = {}
        for flags in (0, capi.KAJO_FLAG_NO_ONE_LIGHT, capi.KAJO_FLAG_NO_ONE_LIGHT | capi.KAJO_FLAG_NO_SPLIT, capi.KAJO_FLAG_NO_SPLIT):
            with HipRenderer(sc, W, H, spp=S, seed=SEED, depth_limit=depth, strict=(numerics == "strict"), exact=(numerics == "exact"), counters=True,
                             flags=flags, passes_per_launch=ppl) as r:
                got[flags] = (r.render(passes).radiance(), r.counters())
        walks = lambda c: c["traversals"] + c["shadowQueries"]
        for flags in got:
            assert bits_equal(got[0][0], got[flags][0]), (key, numerics, flags)
            assert walks(got[0][1]) == walks(got[flags][1]) and got[0][1]["vertices"] == got[flags][1]["vertices"], (key, numerics, flags, got[0][1], got[flags][1])


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_shadow_lists_on_adversarial_geometry(O, scenes, seed):
    """Overlapping and nested spheres, big lights, a light cut by the ceiling plane, a light touching a sphere, 1 cm spheres
    (tests/test_shadow_lists_cpu.py adversarial_scene): STRICT with lists = STRICT through the grid = the oracle's every-object
    walk, bit for bit; FAST with lists = FAST through the grid."""
    from kajo_amd import capi
    from test_shadow_lists_cpu import adversarial_scene
    sc = adversarial_scene(scenes["spheres_a169"], seed)
    W, H, S, passes = 96, 54, 9, 2
    want = O.create(sc, math=1).render(W, H, S=S, passes=passes, seed=SEED, depth_limit=8)
    got = {}
    for strict in (True, False):
        for flags in (0, capi.KAJO_FLAG_NO_SHADOW_LISTS):
            with HipRenderer(sc, W, H, spp=S, seed=SEED, strict=strict, counters=True, flags=flags) as r:
                got[strict, flags] = (r.render(passes).radiance(), r.counters())
    assert got[True, 0][1]["shadowQueries"] > 0 and got[True, capi.KAJO_FLAG_NO_SHADOW_LISTS][1]["shadowQueries"] == 0
    for flags in (0, capi.KAJO_FLAG_NO_SHADOW_LISTS):
        g = got[True, flags][0][..., :3]
        assert ((g.view(np.uint32) == want[..., :3].view(np.uint32)) | (np.isnan(g) & np.isnan(want[..., :3]))).all(), (seed, flags)
    a, b = got[False, 0][0], got[False, capi.KAJO_FLAG_NO_SHADOW_LISTS][0]
    assert ((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all(), seed


def test_strict_passes_split_and_reset(scenes):
    sc = scenes["spheres_a1"]
    with HipRenderer(sc, 40, 24, strict=True, passes_per_launch=1) as a, HipRenderer(sc, 40, 24, strict=True) as b:
        fa = a.render(1).render(2).radiance()
        fb = b.render(3).radiance()
        assert bits_equal(fa, fb)
        a.reset()
        assert a.passes == 0
        assert bits_equal(a.render(3).radiance(), fb)


@pytest.mark.parametrize("count", [2, 3, 8])
def test_tiling_is_bit_invariant(scenes, count):
    """1/2/3/8 tile owners on one GPU, gathered as the RCCL gather would deliver them."""
    import torch
    sc = scenes["spheres_a169"]
    W, H = 200, 70
    for strict in (True, False):
        with HipRenderer(sc, W, H, strict=strict, tile=(32, 8)) as one:
            want = one.render(1).radiance()
        parts = [HipRenderer(sc, W, H, strict=strict, tile=(32, 8), tile_index=i, tile_count=count) for i in range(count)]
        sizes = {p.tile_buffer()[1] for p in parts}
        assert len(sizes) == 1
        nbytes = sizes.pop()
        gathered = torch.empty(count * nbytes // 4, dtype=torch.float32, device="cuda")
        for i, p in enumerate(parts):
            p.render(1).wait()
            ptr, _ = p.tile_buffer()
            hip = C.CDLL("libamdhip64.so")
            rc = hip.hipMemcpy(C.c_void_p(gathered.data_ptr() + i * nbytes), C.c_void_p(ptr), C.c_size_t(nbytes), C.c_int(3))
            assert rc == 0
        parts[0].compose(gathered.data_ptr())
        got = parts[0].radiance()
        assert bits_equal(got, want), (count, strict)
        # the host-side mirror of the tile map (kajo_amd/tiles.py, used for the gloo test) agrees
        from kajo_amd.tiles import TileLayout
        lay = TileLayout(W, H, count, (32, 8))
        assert lay.slots_per_owner * 16 == nbytes
        assert bits_equal(lay.compose(gathered.cpu().view(count, lay.slots_per_owner, 4).numpy()), want)
        for p in parts:
            p.close()


def test_fast_vs_oracle_and_golden(O, golden, scenes):
    z = golden.frames
    frames = json.loads(str(z["frames"]))
    seed = int(z["seed"])
    for name, key, W, H, S, passes, depth in frames:
        sc = scenes[key]
        with HipRenderer(sc, W, H, spp=S, depth_limit=depth, seed=seed) as r:
            acc = r.render(passes).radiance()
            px = r.argb8()
        est = acc[..., :3] / passes
        want = O.create(sc, math=0).render(W, H, S=S, passes=passes, seed=seed, depth_limit=depth)[..., :3] / passes
        s = frame_stats(est, want)
        assert s["median"] <= 1e-5 and s["p99"] <= 2e-3 and s["clamped_rmse"] <= 1e-3, (name, "vs oracle", s)
        for tag in ("strict", "fast"):
            g = frame_stats(est, z["%s/rgb_%s" % (name, tag)] / passes)
            floor = frame_stats(z[name + "/rgb_strict"] / passes, z[name + "/rgb_fast"] / passes)
            assert g["median"] <= max(1e-5, 1.25 * floor["median"]), (name, tag, g, floor)
            assert g["p99"] <= max(2e-3, 1.25 * floor["p99"]), (name, tag, g, floor)
            assert g["clamped_rmse"] <= max(1e-3, 1.25 * floor["clamped_rmse"]), (name, tag, g, floor)
        # 8-bit image against the reference's own resolve of its own frame
        ref_px = z[name + "/argb8_strict"]
        chan = lambda p: np.stack([(p >> 16) & 255, (p >> 8) & 255, p & 255], -1).astype(int)
        diff = np.abs(chan(px) - chan(ref_px))
        assert (px >> 24 == 255).all()
        assert np.mean(diff > 1) <= 0.01, (name, np.mean(diff > 1))


def test_resolve_strict_matches_oracle(O, scenes):
    sc = scenes["spheres_a1"]
    with HipRenderer(sc, 64, 48, strict=True) as r:
        acc = r.render(2).radiance()
        px = r.argb8()
    want = O.resolve(acc, 2, math=1).reshape(48, 64)
    ok = np.isfinite(acc[..., :3]).all(-1)
    assert np.array_equal(px[ok], want[ok])


def test_counters(scenes):
    sc = scenes["spheres_a169"]
    W, H = 192, 108
    with HipRenderer(sc, W, H, counters=True) as r:
        c = r.render(2).counters()
    assert c["passes"] == 2 and c["paths"] == W * H * 25 * 2
    assert c["primitiveTests"] == c["traversals"] * 11
    # T_min = 1.887 traversals / path, 1.35-1.54 vertices / path on spheres.json 16:9
    # (the kernel also skips shadow rays whose BSDF pdf is zero)
    assert 1.3 < c["traversals"] / c["paths"] < 2.1
    assert 1.3 < c["vertices"] / c["paths"] < 1.8
    assert 0.3 < c["traversals"] / c["laneSlots"] <= 1.0
    assert c["kernelMs"] > 0 and c["launches"] == 1


def test_full_size_properties(scenes):
    """BASELINE configs[1] size (1920x1080): size-independent properties instead of the oracle --
    determinism, pass additivity, and agreement of a cropped region with the oracle."""
    sc = scenes["spheres_a169"]
    W, H = 1920, 1080
    with HipRenderer(sc, W, H) as r:
        a = r.render(1).radiance()
        r.reset()
        b = r.render(1).radiance()
        assert bits_equal(a, b), "two runs differ"
        assert np.isfinite(a[..., :3]).mean() > 0.9999
        est = a[..., :3]
        # the emissive sphere must be visible: its radiance/S*25 = 445.72 * 25/32
        assert np.nanmax(est) > 300
    O = OracleLib("oracle")
    x0, y0, w, h = 900, 500, 64, 32
    want = O.create(sc, 0).render(W, H, S=32, passes=1, seed=SEED, rect=(x0, y0, w, h))[y0:y0 + h, x0:x0 + w, :3]
    s = frame_stats(a[y0:y0 + h, x0:x0 + w, :3], want)
    assert s["median"] <= 1e-5 and s["p99"] <= 5e-3, s


@pytest.mark.parametrize("passes", [16, 12, 6, 5])
def test_pass_splitting_over_waves_is_bit_invariant(O, scenes, passes):
    """Small frames: several waves share a pixel block and divide the passes of a launch (up to 16 waves for 16 passes;
    12 -> 4, 6 -> 2, 5 -> none). The per-pass terms are added in pass order, so the buffer is the one a single wave per
    block produces -- for the FAST kernels bit for bit against KAJO_FLAG_NO_SPLIT, for the STRICT ones against the oracle."""
    from kajo_amd import capi
    sc = scenes["spheres_a169"]
    W, H = 72, 40  # 45 pixel blocks, ragged
    with HipRenderer(sc, W, H, passes_per_launch=16) as a, HipRenderer(sc, W, H, passes_per_launch=16, flags=capi.KAJO_FLAG_NO_SPLIT) as b:
        assert bits_equal(a.render(passes).radiance(), b.render(passes).radiance())
        # and across launches: the accumulation is continued
        assert bits_equal(a.render(4).radiance(), b.render(4).radiance())
    want = O.create(sc, 1).render(W, H, S=32, passes=passes, seed=0o715517, depth_limit=8)
    with HipRenderer(sc, W, H, strict=True, passes_per_launch=16) as r:
        assert bits_equal(r.render(passes).radiance(), want)


@pytest.mark.parametrize("S,ppl,passes", [(16, 1, 3), (32, 1, 2), (32, 2, 4), (9, 3, 3), (36, 1, 1)])
def test_sample_splitting_over_waves_is_bit_invariant(O, scenes, S, ppl, passes):
    """Small frames, launches of few passes (BASELINE configs[0]: ONE pass): the waves of a block divide the n*n samples of a
    pass; every path's radiance goes to an LDS table [pass][sample][pixel] and wave 0 forms the sums in sample order
    (Renderer.cpp:66), so the buffer is the one a single wave per block produces -- FAST bit for bit against
    KAJO_FLAG_NO_SPLIT, STRICT against the oracle; the accumulation continues across launches."""
    from kajo_amd import capi
    sc = scenes["spheres_a169"]
    W, H = 72, 40  # 45 pixel blocks, ragged
    with HipRenderer(sc, W, H, spp=S, passes_per_launch=ppl) as a, HipRenderer(sc, W, H, spp=S, passes_per_launch=ppl, flags=capi.KAJO_FLAG_NO_SPLIT) as b:
        assert bits_equal(a.render(passes).radiance(), b.render(passes).radiance())
        assert bits_equal(a.render(1).radiance(), b.render(1).radiance())
    want = O.create(sc, 1).render(W, H, S=S, passes=passes, seed=0o715517, depth_limit=8)
    with HipRenderer(sc, W, H, spp=S, strict=True, passes_per_launch=ppl) as r:
        assert bits_equal(r.render(passes).radiance(), want)


def test_strict_holding_threshold_does_not_change_a_bit():
    """STRICT loop: the light / BSDF sampling blocks run when `thrL` lanes want them or one has waited a trip
    (integrator.inc.hip MODE_HOLD; the product's constants are 28 STRICT / 20 FAST, 1 for large scenes: capi.cpp). Whatever
    the threshold -- every trip, rarely, never without a wait -- every path takes the same decisions and every sum forms in
    the same order: the oracle bit for bit, 1 and 3 lights. The threshold is a constant of libkajo_hip.so; the sweep runs in a
    child process on the tools' twin libkajo_hip_tune.so (same kernel objects; KAJO_THR_L read by its capi.cpp, tuning.h)."""
    import subprocess
    import sys
    import textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tune = os.path.join(root, "kajo_amd", "libkajo_hip_tune.so")
    if not os.path.exists(tune):
        pytest.skip("libkajo_hip_tune.so not built (make -C kajo_amd/csrc tune)")
    code = textwrap.dedent("""
        import os, sys
        sys.path[:0] = [%r, %r]
        import numpy as np
        from kajo_amd import capi
        from kajo_amd.renderer import HipRenderer
        from kajo_amd.scene import Scene
        from oraclelib import OracleLib
        from test_hip_parity import bits_equal
        assert capi.LIB_PATH.endswith("libkajo_hip_tune.so")
        z = np.load(%r)
        O = OracleLib("oracle")
        for key, W, H, passes in (("spheres_a169", 96, 54, 3), ("caustics_a169", 80, 45, 2)):
            sc = Scene.from_npz(z, key + "/", key)
            want = O.create(sc, 1).render(W, H, S=32, passes=passes, seed=0o715517, depth_limit=8)
            for thr in (1, 8, 40, 64):
                os.environ["KAJO_THR_L"] = str(thr)
                with HipRenderer(sc, W, H, strict=True, passes_per_launch=2) as r:
                    assert bits_equal(r.render(passes).radiance(), want), (key, thr)
        print("hold ok")
    """) % (root, os.path.join(root, "tests"), os.path.join(root, "tests", "golden", "scenes.npz"))
    p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, KAJO_HIP_LIB=tune), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "hold ok" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]
