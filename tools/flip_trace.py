#!/usr/bin/env python3
"""Which paths make the FAST kernels differ from the oracle, and at which vertex do they part?

Renders spheres.json 256x144, 16 passes with the FAST kernels and the oracle(libm), takes the worst pixels, replays
every one of their 400 camera paths singly (oracle camera ray + generator state -> kajo_hip_kat_shade FAST and STRICT,
koracle_shade), and for a path whose FAST radiance differs finds the first depth limit at which it does: that is the
vertex where the two took different decisions. Prints the oracle's event log of the path up to there."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch  # noqa: F401
from kajo_amd.renderer import HipRenderer
from kajo_amd.scene import Scene
from oraclelib import OracleLib, debug_path, camera_ray, _p

W, H, S, PASSES, SEED = 256, 144, 32, 16, 0o715517
top = int(sys.argv[1]) if len(sys.argv) > 1 else 12
z = np.load(os.path.join(ROOT, "tests", "golden", "scenes.npz"))
scene = Scene.from_npz(z, "spheres_a169/", "spheres")
orc = OracleLib("oracle").create(scene, 0)
want = orc.render(W, H, S=S, passes=PASSES, seed=SEED, depth_limit=8, threads=64)[..., :3] / PASSES
r = HipRenderer(scene, W, H, spp=S, depth_limit=8, seed=SEED)
got = r.render(PASSES).radiance()[..., :3] / PASSES
r.close()
d = np.abs(np.clip(got, 0, 1) - np.clip(want, 0, 1)).max(-1)
order = np.argsort(d.ravel())[::-1][:top]

def camera_rays(x, y):
    rays = np.zeros((PASSES * 25, 6), np.float32)
    states = np.zeros((PASSES * 25, 2), np.uint64)
    k = 0
    for p in range(1, PASSES + 1):
        for s in range(25):
            o, d, st = camera_ray(orc, W, H, S, x, y, s, npass=p, seed=SEED)
            rays[k, :3], rays[k, 3:], states[k] = o, d, st
            k += 1
    return rays, states

kinds = {1: "vertex id/depth/t", 2: "lobe kind/s", 3: "light sample pb/pl/Ld.x", 4: "extension p/pL/wb.x", 5: "  position", 6: "  normal",
         7: "  light direction", 8: "  next origin", 9: "  next direction"}
mat_names = None
summary = []
for i in order:
    y, x = divmod(int(i), W)
    rays, states = camera_rays(x, y)
    per = {}
    for limit in range(0, 9):
        fast = HipRenderer(scene, W, H, spp=S, depth_limit=limit, seed=SEED)
        f_rgb, _ = fast.kat_shade(rays[:, :3], rays[:, 3:], states)
        fast.close()
        o_rgb, _ = orc.shade(rays[:, :3], rays[:, 3:], states, depth_limit=limit)
        per[limit] = (f_rgb, o_rgb)
    f8, o8 = per[8]
    diff = np.abs(f8 - o8).max(-1)
    rel = diff / np.maximum(np.abs(o8).max(-1), 1e-3)
    bad = np.nonzero(rel > 1e-3)[0]
    print("pixel (%d,%d): |clamped delta| %.2e; estimate FAST %s oracle %s; sum over replayed paths/400: FAST %s oracle %s; %d of 400 paths differ > 1e-3 rel"
          % (x, y, d[y, x], got[y, x], want[y, x], f8.sum(0) / 400 * 25 / S, o8.sum(0) / 400 * 25 / S, bad.size))
    for k in bad[:3]:
        first = next((lim for lim in range(9) if np.abs(per[lim][0][k] - per[lim][1][k]).max() > 1e-3 * max(np.abs(per[lim][1][k]).max(), 1e-3)), None)
        p, s = k // 25 + 1, k % 25
        log, rgb = debug_path(orc, W, H, S, x, y, s, npass=p, seed=SEED, depth_limit=8)
        print("   path pass %d sample %d: FAST %s oracle %s; first differing depth limit: %s" % (p, s, f8[k], o8[k], first))
        for rec in log:
            code = int(rec[0])
            print("      %-26s %s" % (kinds.get(code, str(code)), " ".join("%.7g" % v for v in rec[1:])))
        # the FAST closest-hit walk on the oracle's own segments: does any single segment explain the difference?
        segO = [rays[k, :3]] + [rec[1:] for rec in log if int(rec[0]) == 8]
        segD = [rays[k, 3:]] + [rec[1:] for rec in log if int(rec[0]) == 9]
        vpos = [rec[1:] for rec in log if int(rec[0]) == 5]
        vnor = [rec[1:] for rec in log if int(rec[0]) == 6]
        vid = [int(rec[1]) for rec in log if int(rec[0]) == 1]
        m = min(len(segO), len(vpos))
        fast = HipRenderer(scene, W, H, spp=S, depth_limit=8, seed=SEED)
        tr = fast.kat_trace(np.array(segO[:m], np.float32), np.array(segD[:m], np.float32))
        fast.close()
        for j in range(m):
            print("      segment %d: FAST id %d (oracle %d), |dP| %.2e, |dN| %.2e" % (
                j, tr["idx"][j], vid[j], np.abs(tr["position"][j] - vpos[j]).max(), np.abs(tr["normal"][j] - vnor[j]).max()))
        summary.append((x, y, p, s, first, [tuple(rec) for rec in log if int(rec[0]) == 1]))
print()
print("vertex sequences (object id at each depth) of the differing paths; ids: 1..%d planes, then spheres" % scene.n_planes)
for x, y, p, s, first, verts in summary:
    print("  (%3d,%3d) pass %2d sample %2d first-diff depth %s: %s" % (x, y, p, s, first, " -> ".join("%d" % v[1] for v in verts)))
