#!/usr/bin/env python3
"""Round-4 additions to the golden frames: crops of BASELINE configs[2], [3] and [4] at their own frame sizes, rendered by the
COMPILED REFERENCE (oracle/_ref through oracle/ref_harness.cpp: renderer/cpu/Renderer.cpp:36-75 on the reference's own
Raytracer / Shader / BSDF / Light / Random objects, both builds), written to tests/golden/frames3.npz. With them no config is
compared with this repo's oracle alone, whether or not the built reference libraries travelled with a snapshot.

  c3_4k         configs[2]: spheres.json 16:9, 3840 x 2160, all 64 passes x S = 32, depth 8: eight 64 x 32 crops
  c4_1080p      configs[3]: the caustics scene (ideal-reflector wall, glass, Phong, 3 lights; BSDF.cpp:76-96), 1920 x 1080,
                16 passes: ten crops (incl. the mirror wall)
  c4_1080p_128  the same scene at its own 4096 spp = 128 passes: three crops
  c5_4k         configs[4]: 1000 spheres / 16 lights (kajo_amd.scene.stress_scene, seeded), 3840 x 2160, 2 passes: eight crops
  c5_4k_32      the same scene at its own 1024 spp = 32 passes: two crops

Per entry: `crops` (x, y, w, h), `crop_names`, `rgb_crops_strict` / `rgb_crops_fast` (float sums over the passes, NOT divided),
`scene_crc` (CRC-32 of the scene's sphere and plane arrays: a changed generator fails the test instead of comparing other scenes).

Run in the build container only (needs /root/reference and oracle/_ref):  python tests/golden/make_golden_frames3.py
The file holds data only: crop rectangles and the reference's outputs.
"""
import json
import os
import sys
import time
import zlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from kajo_amd.scene import Scene, stress_scene  # noqa: E402
from oraclelib import OracleLib  # noqa: E402
from workload_crops import crops_for  # noqa: E402

SEED = 0o715517
THREADS = max(1, min(8, os.cpu_count() or 1))


def scene_crc(sc):
    return np.uint32(zlib.crc32(np.ascontiguousarray(sc.planes).tobytes(), zlib.crc32(np.ascontiguousarray(sc.spheres).tobytes())))


def entry(out, key, libs, sc, W, H, passes, crops):
    out[key + "/crops"] = np.array([(x, y, w, h) for _, x, y, w, h in crops], np.int32)
    out[key + "/crop_names"] = json.dumps([c[0] for c in crops])
    out[key + "/scene_crc"] = scene_crc(sc)
    out[key + "/passes"] = np.int32(passes)

    def job(a):
        tag, (name, x, y, w, h) = a
        q = libs[tag].create(sc)  # one handle per job: the harness is single-threaded, ctypes releases the GIL
        acc = q.render(W, H, S=32, passes=passes, seed=SEED, depth_limit=8, rect=(x, y, w, h))[y:y + h, x:x + w, :3].copy()
        q.close()
        return acc

    t0 = time.time()
    with ThreadPoolExecutor(THREADS) as ex:
        for tag in ("strict", "fast"):
            out[key + "/rgb_crops_" + tag] = np.stack(list(ex.map(job, [(tag, c) for c in crops])))
    print("%-14s %d crops x %d passes: %.1f s" % (key, len(crops), passes, time.time() - t0), flush=True)


def main():
    libs = {"fast": OracleLib("ref"), "strict": OracleLib("ref_strict")}
    z = np.load(os.path.join(HERE, "scenes.npz"))
    a169 = Scene.from_npz(z, "spheres_a169/", "spheres_a169")
    caustics = Scene.from_npz(z, "caustics_a169/", "caustics_a169")
    stress = stress_scene(a169, 1000, 16)
    out = {"seed": np.uint64(SEED)}
    entry(out, "c3_4k", libs, a169, 3840, 2160, 64, crops_for(a169, 3840, 2160, 8))
    entry(out, "c4_1080p", libs, caustics, 1920, 1080, 16, crops_for(caustics, 1920, 1080, 10))
    entry(out, "c4_1080p_128", libs, caustics, 1920, 1080, 128, crops_for(caustics, 1920, 1080, 10)[:3])
    entry(out, "c5_4k", libs, stress, 3840, 2160, 2, crops_for(stress, 3840, 2160, 8))
    entry(out, "c5_4k_32", libs, stress, 3840, 2160, 32, crops_for(stress, 3840, 2160, 10)[:2])
    path = os.path.join(HERE, "frames3.npz")
    np.savez_compressed(path, **out)
    print("frames3.npz %d bytes" % os.path.getsize(path))


if __name__ == "__main__":
    main()
