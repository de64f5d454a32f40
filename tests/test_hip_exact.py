"""The EXACT ("decision-exact") numerics build, KAJO_FLAG_EXACT (kernel_exact.hip; integrator.inc.hip KAJO_RSTRICT).

Claim under test: a path rendered by the EXACT kernels takes every DECISION of the oracle's path -- it meets the same objects, draws
the same random numbers (Random.cpp:27-53), ends on the same vertex in the same generator state -- and differs from it only in the
last places of the products that scale its radiance (BSDF values and pdfs, BSDF.cpp:30-39,62-74,87-91; the light pdf, Light.cpp:48-62;
the MIS weight and throughput, Shader.cpp:74-83,203-212). So:
  * known-answer paths: final generator states EQUAL the oracle's on every path; radiance within 2e-4 relative;
  * frames: the same pixels are NaN; every other pixel within 3e-4 relative of the oracle's; clamped RMSE far below BASELINE.json's 1e-4.
Everything goes through the C ABI."""
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


def compare(got, want, passes, what, max_rel=3e-4, rmse=1e-6):
    """got, want: (H, W, >=3) sums over passes. -> dict of figures; asserts the decision-exact bounds."""
    g, w = got[..., :3].astype(np.float64) / passes, want[..., :3].astype(np.float64) / passes
    nan_g, nan_w = ~np.isfinite(g).all(-1), ~np.isfinite(w).all(-1)
    assert np.array_equal(nan_g, nan_w), "%s: not-a-number pixels differ: %d here, %d in the oracle" % (what, nan_g.sum(), nan_w.sum())
    m = ~nan_w
    d = np.abs(g - w)[m]
    rel = d / np.maximum(np.abs(w[m]), 1e-3)
    rmse = float(np.sqrt(np.mean((np.clip(g[m], 0, 1) - np.clip(w[m], 0, 1)) ** 2)))
    s = dict(max_abs=float(d.max()), max_rel=float(rel.max()), clamped_rmse=rmse, identical=float(np.mean(d == 0)), nan=int(nan_w.sum()))
    # a path that decided differently would move its pixel by one path's radiance / (n^2 passes): 1e-3 and more
    assert s["max_rel"] <= max_rel and s["clamped_rmse"] <= rmse, (what, s)
    return s


@pytest.mark.parametrize("key", ["spheres_a1", "test_a1"])
@pytest.mark.parametrize("depth", [0, 1, 8])
def test_exact_paths_end_in_the_oracles_generator_state(O, golden, scenes, key, depth):
    """kajo_hip_kat_shade: one path per given (ray, 128-bit generator state). The state a path ENDS in pins every draw it made --
    coins, light samples, BSDF samples -- and with them every decision that depends on a hit: EXACT == oracle on every path."""
    z = golden.kat_shade
    o, d, st = z[key + "/origins"], z[key + "/dirs"], z[key + "/states"]
    want, wfin = O.create(scenes[key], 1).shade(o, d, st, depth)
    with HipRenderer(scenes[key], 8, 8, depth_limit=depth, exact=True) as r:
        rgb, fin = r.kat_shade(o, d, st)
    assert np.array_equal(fin, wfin), "%d of %d paths end in another state" % ((fin != wfin).any(1).sum(), len(fin))
    nan_g, nan_w = np.isnan(rgb).any(1), np.isnan(want).any(1)
    assert np.array_equal(nan_g, nan_w)
    ok = ~nan_w
    rel = np.abs(rgb - want)[ok].max(1) / np.maximum(np.abs(want[ok]).max(1), 1e-6)
    # (a Phong lobe of exponent 100 carries the last places of its cosine a hundredfold; a path that DECIDED differently is off by O(1))
    # (measured: 7-8e-5 at most, 2-11 % of the paths beyond 2e-5 -- the Phong vertices: the oracle's weight value / pdf there is
    # pow(R.d, e) / pow(cos a, e) with R.d the re-derived cosine of the sampled angle a, BSDF.cpp:55-74, where EXACT has their quotient 1)
    assert rel.max() <= 2e-4, rel.max()
    # ... and the STRICT kernels' states, for the record of what is being matched
    with HipRenderer(scenes[key], 8, 8, depth_limit=depth, strict=True) as r:
        _, sfin = r.kat_shade(o, d, st)
    assert np.array_equal(fin, sfin)


CASES = [
    # scene key, W, H, S, passes, depth, passes per launch
    ("spheres_a1", 256, 256, 16, 1, 1, 0),   # BASELINE configs[0]: the sample-splitting launch
    ("spheres_a169", 256, 144, 32, 16, 8, 0),  # configs[1] in small: the one-light instance, 16 passes in one launch
    ("spheres_a169", 100, 37, 32, 3, 8, 2),  # ragged, two launches
    ("test_a1", 96, 96, 32, 4, 8, 0),        # second material mix (test.json)
    ("spheres_a1", 33, 1, 4, 1, 0, 0),       # one row, depth limit 0
]


@pytest.mark.parametrize("key,W,H,S,passes,depth,ppl", CASES)
def test_exact_frames_against_the_oracle(O, scenes, key, W, H, S, passes, depth, ppl):
    sc = scenes[key]
    want = O.create(sc, math=1).render(W, H, S=S, passes=passes, seed=SEED, depth_limit=depth)
    with HipRenderer(sc, W, H, spp=S, depth_limit=depth, seed=SEED, exact=True, passes_per_launch=ppl) as r:
        got = r.render(passes).radiance()
    compare(got, want, passes, key)


def test_exact_frames_of_the_generated_scenes(O, scenes):
    """Three lights (shadow walks inside the light loop), the mirror wall, glass; 300 spheres through the uniform grid; 1000 spheres
    / 16 lights through grid + visibility lists: the large-scene instances of the EXACT build."""
    base = scenes["spheres_a169"]
    for sc, W, H, S, passes in ((caustics_scene(base), 128, 72, 32, 4), (stress_scene(base, 300, 6, seed=7), 96, 54, 16, 2),
                                (stress_scene(base, 1000, 16), 64, 36, 16, 2)):
        want = O.create(sc, math=1).render(W, H, S=S, passes=passes, seed=SEED, depth_limit=8)
        with HipRenderer(sc, W, H, spp=S, seed=SEED, exact=True) as r:
            got = r.render(passes).radiance()
        # (half of these scenes' spheres are Phong lobes of exponent 10 / 50 / 100, and 32 paths make a pixel: the oracle's value / pdf
        # quotient pow(R.d, e) / pow(cos a, e) carries the last places of the two cosines a hundredfold per bounce, where EXACT has 1 --
        # measured 2.8e-4 relative on the worst pixel, clamped RMSE 1.4e-6)
        compare(got, want, passes, sc.name, max_rel=2e-3, rmse=5e-6)


def test_exact_and_strict_flags_exclude_each_other(scenes):
    from kajo_amd import capi
    with pytest.raises(capi.KajoError):
        HipRenderer(scenes["spheres_a1"], 8, 8, strict=True, exact=True)


def test_exact_image_is_the_strict_image(O, scenes):
    """The 8-bit image (Renderer.cpp:73-75, Image.cpp:14-27) of the two builds: radiance differences in the seventh digit move a
    channel only where the value sits on a rounding boundary of c * 255 + .5."""
    sc = scenes["spheres_a169"]
    W, H = 256, 144
    with HipRenderer(sc, W, H, seed=SEED, exact=True) as r:
        a = r.render(4).argb8()
    with HipRenderer(sc, W, H, seed=SEED, strict=True) as r:
        b = r.render(4).argb8()
    ch = lambda x: np.stack([(x >> s) & 255 for s in (16, 8, 0)], -1).astype(np.int32)
    diff = np.abs(ch(a) - ch(b))
    assert diff.max() <= 1 and np.mean(diff != 0) <= 5e-4, (diff.max(), np.mean(diff != 0))  # measured 1.8e-4


def test_division_and_square_root_formed_by_hand_are_ieee(scenes):
    """integrator.inc.hip kdiv / ksqrt: hipcc's own IEEE sequences without their range scaling (9 instructions each instead of 11 /
    16). Inside the stated range -- operands in 2^-47 .. 2^47, zeros, infinities, NaNs -- every quotient and root must carry the bits of
    the IEEE operation (numpy's, i.e. the x86 instruction the oracle executes, Raytracer.cpp:30-44)."""
    import ctypes as C
    from kajo_amd import capi
    rng = np.random.default_rng(5)
    n = 1 << 22
    def operands():
        m = rng.integers(0, 1 << 23, n, dtype=np.uint32)
        e = rng.integers(127 - 47, 127 + 47, n, dtype=np.uint32)
        sgn = rng.integers(0, 2, n, dtype=np.uint32) << 31
        return (sgn | (e << 23) | m).view(np.float32)
    a, b = operands(), operands()
    # near-ties of the rounding: quotients of operands with short mantissas, perfect squares and their neighbours
    a[:4096] = rng.integers(1, 1 << 12, 4096).astype(np.float32)
    b[:4096] = rng.integers(1, 1 << 12, 4096).astype(np.float32)
    sq = (rng.integers(1, 1 << 12, 4096).astype(np.float32)) ** 2
    special = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1.0, -1.0, 3.0, 0.1, 1e-12, 1e12], np.float32)
    sa, sb = np.meshgrid(special, special)
    a[4096:4096 + sa.size], b[4096:4096 + sb.size] = sa.ravel(), sb.ravel()
    p = lambda v: v.ctypes.data_as(C.c_void_p)
    with HipRenderer(scenes["spheres_a1"], 8, 8, strict=True) as r:
        out = np.zeros(n, np.float32)
        capi.check(capi.lib().kajo_hip_kat_strictmath(r._h, 5, n, p(a), p(b), p(out)))
        with np.errstate(all="ignore"):
            want = a / b
        same = (out.view(np.uint32) == want.view(np.uint32)) | (np.isnan(out) & np.isnan(want))
        assert same.all(), (int((~same).sum()), a[~same][:4], b[~same][:4], out[~same][:4], want[~same][:4])
        x = np.abs(a)
        x[:4096] = sq
        x[4096:8192] = np.nextafter(sq, np.float32(np.inf))
        x[8192:12288] = np.nextafter(sq, np.float32(0))
        x[12288:12288 + special.size] = special
        capi.check(capi.lib().kajo_hip_kat_strictmath(r._h, 6, n, p(x), p(x), p(out)))
        with np.errstate(all="ignore"):
            want = np.sqrt(x)
        same = (out.view(np.uint32) == want.view(np.uint32)) | (np.isnan(out) & np.isnan(want))
        assert same.all(), (int((~same).sum()), x[~same][:4], out[~same][:4], want[~same][:4])
