"""One-off soak of the large-scene kernels (pair loop, visibility lists, grid): seeded draws of scenes with 48-500 spheres and 1-40 lights, ragged frames,
pass splits, depth limits. STRICT must equal the oracle's every-object walk bit for bit; EXACT must have its not-a-number pixels and stay within rounding;
FAST with lists must equal FAST through the grid. `kind=small`: the small-scene kernels instead, on the generated rooms of rotated spheres of every
material with 1-4 lights (tests/scenes_extra.py). usage: large_scene_soak.py [draws=30] [seed=1] [kind=large|small]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, warnings
warnings.filterwarnings("ignore")
from kajo_amd import capi
from kajo_amd.renderer import HipRenderer
from kajo_amd.scene import Scene, stress_scene
from oraclelib import OracleLib
from test_shadow_lists_cpu import adversarial_scene
kw = dict(a.split("=") for a in sys.argv[1:])
draws, seed0, kind = int(kw.get("draws", 30)), int(kw.get("seed", 1)), kw.get("kind", "large")
from scenes_extra import mixed_scene
z = np.load(os.path.join(ROOT, "tests", "golden", "scenes.npz"))
base = Scene.from_npz(z, "spheres_a169/", "s")
lib = OracleLib("oracle")
rng = np.random.default_rng(seed0)
bad = 0
for draw in range(draws):
    nl = int(rng.choice([1, 2, 3, 5, 7, 8, 15, 16, 17, 24, 33, 40]))
    if kind == "small":
        sc = mixed_scene(base, int(rng.integers(1, 1000000)))
    elif draw % 4 == 3:
        sc = adversarial_scene(base, int(rng.integers(10, 100000)), n=int(rng.integers(60, 200)), n_lights=min(nl, 12))
    else:
        sc = stress_scene(base, int(rng.integers(48, 500)), nl, seed=int(rng.integers(1, 100000)))
    W, H = int(rng.choice([33, 64, 97, 128, 160])), int(rng.choice([17, 40, 54, 72]))
    S, passes, depth = int(rng.choice([4, 9, 16, 25])), int(rng.integers(1, 5)), int(rng.integers(1, 9))
    ppl, seed = int(rng.choice([0, 1, 2])), int(rng.integers(1, 2 ** 40))
    t0 = time.time()
    want = lib.create(sc, 1).render(W, H, S=S, passes=passes, seed=seed, depth_limit=depth, threads=16)
    msg = []
    for flags in (0, capi.KAJO_FLAG_NO_SHADOW_LISTS):
        with HipRenderer(sc, W, H, spp=S, depth_limit=depth, seed=seed, strict=True, passes_per_launch=ppl, flags=flags) as r:
            got = r.render(passes).radiance()
        same = ((got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want)))[..., :3]
        if not same.all():
            msg.append("STRICT flags=%d: %d channels differ" % (flags, int((~same).sum())))
    ex = []
    for flags in (0, capi.KAJO_FLAG_NO_SHADOW_LISTS):
        with HipRenderer(sc, W, H, spp=S, depth_limit=depth, seed=seed, exact=True, passes_per_launch=ppl, flags=flags) as r:
            ex.append(r.render(passes).radiance()[..., :3])
    if not ((ex[0].view(np.uint32) == ex[1].view(np.uint32)) | (np.isnan(ex[0]) & np.isnan(ex[1]))).all():
        msg.append("EXACT lists != EXACT grid")
    fin = np.isfinite(want[..., :3]).all(-1)
    if not np.array_equal(np.isfinite(ex[0]).all(-1), fin):
        msg.append("EXACT NaN pixels differ")
    elif fin.any():
        rel = (np.abs(ex[0] - want[..., :3])[fin] / np.maximum(np.abs(want[..., :3][fin]), 1e-3)).max()
        if rel > 2e-3:
            msg.append("EXACT max rel %.2e" % rel)
    fa = []
    for flags in (0, capi.KAJO_FLAG_NO_SHADOW_LISTS):
        with HipRenderer(sc, W, H, spp=S, depth_limit=depth, seed=seed, passes_per_launch=ppl, flags=flags) as r:
            fa.append(r.render(passes).radiance())
    if not ((fa[0].view(np.uint32) == fa[1].view(np.uint32)) | (np.isnan(fa[0]) & np.isnan(fa[1]))).all():
        msg.append("FAST lists != FAST grid")
    bad += bool(msg)
    print("draw %2d: %-16s %3d lights %dx%d S=%d passes=%d depth=%d ppl=%d  %s  (%.0f s)" % (draw, sc.name, sc.n_lights, W, H, S, passes, depth, ppl, "; ".join(msg) or "ok", time.time() - t0), flush=True)
print("%d of %d draws with differences" % (bad, draws))
sys.exit(1 if bad else 0)
