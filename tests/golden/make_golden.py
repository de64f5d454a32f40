#!/usr/bin/env python3
"""Generate the golden known-answer vectors in tests/golden/ from the COMPILED REFERENCE.

Run in the build container only (needs /root/reference and oracle/_ref, built by
`make -C oracle ref`):

    python tests/golden/make_golden.py

Every expected value below is produced by the reference's own objects (scene::Parser,
cpu::Random, cpu::Raytracer::trace, the cpu BSDF / SphericalLight classes, cpu::Shader::shade,
Image::linearToSRGB/colorToRGBA8) through oracle/ref_harness.cpp, twice: from the build with
the reference's flags ("fast": -O3 -ffast-math, keys *_fast) and from an -O2 build of the same
sources ("strict", keys *_strict). The pair gives the reference-vs-reference tolerance floor that
travels with the fixtures (fixture 8). Inputs are seeded numpy draws.

The files hold data only: inputs and the reference's outputs.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oraclelib import OracleLib  # noqa: E402

REF = "/root/reference"
SEED = 0o715517  # 236367, cpu/Random.h:43


def unit(v):
    return v / np.linalg.norm(v, axis=1, keepdims=True)


def make_rays(basis, n, rng):
    """Half camera rays (through uniformly random image points), half interior rays."""
    p1, p2, p3, origin = basis.astype(np.float64)
    n_cam = n // 2
    sx = rng.random(n_cam)
    sy = rng.random(n_cam)
    d = p1 + np.outer(sx, p2 - p1) + np.outer(sy, p3 - p1) - origin
    cam_o = np.repeat(origin[None], n_cam, 0)
    cam_d = unit(d)
    n_in = n - n_cam
    # room interior of data/spheres.json (world is Y-down): x[-8,10] y[-2,1] z[-2,6]
    o = np.stack([rng.uniform(-7.5, 9.5, n_in), rng.uniform(-1.9, .9, n_in), rng.uniform(-1.9, 5.9, n_in)], 1)
    dd = unit(rng.normal(size=(n_in, 3)))
    return (np.concatenate([cam_o, o]).astype(np.float32), np.concatenate([cam_d, dd]).astype(np.float32))


def make_states(n, rng):
    return rng.integers(0, 2 ** 64, size=(n, 2), dtype=np.uint64)


def main():
    fast = OracleLib("ref")
    strict = OracleLib("ref_strict")
    rng = np.random.default_rng(20121012)

    # ---- (1) parsed scenes -------------------------------------------------------------
    scenes = {}
    # the reference's own scene files, and this repo's scene files (kajo_amd/data) parsed BY THE
    # REFERENCE'S PARSER: the expected output of the host-side loader (kajo_amd/host/scene/SceneLoader.cpp)
    own = os.path.join(ROOT, "kajo_amd", "data")
    cases = [("spheres_a1", os.path.join(REF, "data/spheres.json"), 1.0),
             ("spheres_a169", os.path.join(REF, "data/spheres.json"), 1920.0 / 1080.0),
             ("spheres_a43", os.path.join(REF, "data/spheres.json"), 640.0 / 480.0),
             ("test_a1", os.path.join(REF, "data/test.json"), 1.0),
             ("caustics_a169", os.path.join(own, "caustics.json"), 1920.0 / 1080.0),
             ("dialect_a1", os.path.join(own, "dialect.json"), 1.0)]
    out = {}
    for key, path, aspect in cases:
        h = fast.create_from_file(path, aspect)
        sc = h.export_scene(key)
        scenes[key] = sc
        out.update(sc.to_npz_dict(key + "/"))
        h.close()
        h = strict.create_from_file(path, aspect)
        out.update(h.export_scene(key).to_npz_dict(key + "/strict_"))
        h.close()
    out["meta"] = json.dumps({"cases": [[k, os.path.basename(r), a] for k, r, a in cases],
                              "source": "scene::Parser::load (scene/Parser.cpp:214-232), fast build"})
    np.savez_compressed(os.path.join(HERE, "scenes.npz"), **out)

    # ---- (2,3) basis, staging, RNG -----------------------------------------------------
    out = {}
    for key, sc in scenes.items():
        for tag, L in (("fast", fast), ("strict", strict)):
            h = L.create(sc)
            out["%s/basis_%s" % (key, tag)] = h.camera_basis()
            out["%s/staged_%s" % (key, tag)] = h.staged(sc.n_planes + sc.n_spheres)
            h.close()
    seeds = np.array([SEED, 0, 1, 0xdeadbeef], np.uint32)
    out["rng/seeds"] = seeds
    for i, s in enumerate(seeds):
        d, st = fast.rng_from_seed(int(s), 64)
        out["rng/seed%d_draws" % i] = d
        out["rng/seed%d_final" % i] = st
    states = make_states(4, rng)
    out["rng/states"] = states
    for i in range(4):
        d, st = fast.rng_from_state(states[i], 64)
        out["rng/state%d_draws" % i] = d
        out["rng/state%d_final" % i] = st
    coin_states = make_states(256, rng)
    coin_p = rng.random(256).astype(np.float32)
    coin_p[:8] = [0, 1, .5, 1e-8, .999999, .25084, .409826, .859174]
    vals = np.zeros(256, np.int32)
    probs = np.zeros(256, np.float32)
    for i in range(256):
        v, p = fast.flip_coin(coin_states[i], float(coin_p[i]))
        vals[i], probs[i] = v, p
    out.update({"coin/states": coin_states, "coin/p": coin_p, "coin/value": vals, "coin/probability": probs})
    np.savez_compressed(os.path.join(HERE, "kat_basic.npz"), **out)

    # ---- (4) trace ---------------------------------------------------------------------
    out = {}
    for key in ("spheres_a1", "test_a1"):
        sc = scenes[key]
        hf, hs = fast.create(sc), strict.create(sc)
        o, d = make_rays(hf.camera_basis(), 1024, rng)
        out[key + "/origins"], out[key + "/dirs"] = o, d
        for tag, h in (("fast", hf), ("strict", hs)):
            r = h.trace(o, d)
            for k, v in r.items():
                out["%s/%s_%s" % (key, k, tag)] = v
    np.savez_compressed(os.path.join(HERE, "kat_trace.npz"), **out)

    # ---- (5) BSDF / light samples ------------------------------------------------------
    out = {}
    sc = scenes["spheres_a1"]
    hf, hs = fast.create(sc), strict.create(sc)
    o, d = make_rays(hf.camera_basis(), 256, rng)
    st = make_states(256, rng)
    out["origins"], out["dirs"], out["states"] = o, d, st
    kinds = [("lambert", 0, [.25, .5, .75, 1], 0.0, 0), ("phong100", 1, [.4098, .0119, .0119, 1], 100.0, 0),
             ("phong5", 1, [.5, .5, .5, 1], 5.0, 0), ("mirror", 2, [.4098, .4098, .4098, 1], 0.0, 0),
             ("glass2", 3, [.25084, .25084, .25084, 1], 2.0, 0), ("glass1p5", 3, [.9, .9, .9, 1], 1.5, 0),
             ("light4", 4, [0, 0, 0, 0], 0.0, 4)]
    out["kinds"] = json.dumps(kinds)
    for name, kind, color, param, ls in kinds:
        for tag, h in (("fast", hf), ("strict", hs)):
            r = h.sample(kind, o, d, st, color, param, ls)
            for k, v in r.items():
                out["%s/%s_%s" % (name, k, tag)] = v
    np.savez_compressed(os.path.join(HERE, "kat_sample.npz"), **out)

    # ---- (6) shade ---------------------------------------------------------------------
    out = {}
    for key in ("spheres_a1", "test_a1"):
        sc = scenes[key]
        hf, hs = fast.create(sc), strict.create(sc)
        o, d = make_rays(hf.camera_basis(), 1024, rng)
        st = make_states(1024, rng)
        out[key + "/origins"], out[key + "/dirs"], out[key + "/states"] = o, d, st
        for depth in (0, 1, 8):
            for tag, h in (("fast", hf), ("strict", hs)):
                rgb, fin = h.shade(o, d, st, depth)
                out["%s/rgb_d%d_%s" % (key, depth, tag)] = rgb
                out["%s/final_d%d_%s" % (key, depth, tag)] = fin
    np.savez_compressed(os.path.join(HERE, "kat_shade.npz"), **out)

    # ---- (7,8) frames ------------------------------------------------------------------
    out = {}
    frames = [
        # name, scene, W, H, S, passes, depth
        ("c1_64", "spheres_a1", 64, 64, 16, 1, 1),          # BASELINE configs[0] at 64x64
        ("native_64", "spheres_a1", 64, 64, 32, 1, 8),      # reference-native S=32, depth 8
        ("conv_48", "spheres_a1", 48, 48, 32, 16, 8),       # 16 passes
        ("wide_96x54", "spheres_a169", 96, 54, 32, 2, 8),   # 16:9, two passes
        ("test_48", "test_a1", 48, 48, 32, 2, 8),           # data/test.json
    ]
    out["frames"] = json.dumps(frames)
    out["seed"] = np.uint64(SEED)
    for name, key, W, H, S, passes, depth in frames:
        sc = scenes[key]
        for tag, L in (("fast", fast), ("strict", strict)):
            h = L.create(sc)
            acc = h.render(W, H, S=S, passes=passes, seed=SEED, depth_limit=depth)
            out["%s/rgb_%s" % (name, tag)] = acc[..., :3].copy()
            out["%s/argb8_%s" % (name, tag)] = L.resolve(acc, passes).reshape(H, W)
            h.close()
    np.savez_compressed(os.path.join(HERE, "frames.npz"), **out)

    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print("%-16s %8d bytes" % (f, os.path.getsize(os.path.join(HERE, f))))


if __name__ == "__main__":
    main()
