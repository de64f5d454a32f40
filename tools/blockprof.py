"""In-kernel block profile of the render kernel (diagnostic build: `make -C kajo_amd/csrc prof`, run with
KAJO_HIP_LIB=kajo_amd/libkajo_hip_prof.so). Shares of a loop trip per block and the lanes active in each.
usage: blockprof.py [fast|strict|exact] [spheres|caustics|stress] [W H passes]"""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("KAJO_HIP_LIB", os.path.join(ROOT, "kajo_amd", "libkajo_hip_prof.so"))
if not os.path.exists(os.environ["KAJO_HIP_LIB"]):
    sys.exit("blockprof: %s missing -- build it first: make -C kajo_amd/csrc prof" % os.environ["KAJO_HIP_LIB"])
import numpy as np, warnings
warnings.filterwarnings('ignore')
from kajo_amd.renderer import HipRenderer
from kajo_amd.scene import Scene, stress_scene
from kajo_amd import capi

DEFERRED = 'exp' in os.path.basename(os.environ["KAJO_HIP_LIB"]) and os.environ.get('KAJO_BLOCKPROF_DEFERRED', '1') == '1'
if DEFERRED:  # the deferred-shading experiment (deferred.inc.hip): its own blocks and stamp order
    names = ['camera (issue)', 'resume parked vertex', 'shadow ray generated', 'BSDF-sample', 'vertex', 'transparent', 'park (push)', 'shadow-result']
    stamps = ['camera-ray block (before the ray)', 'light+BSDF block (before the ray)', 'traversal', 'vertex/shadow-result block', 'retire/back-edge']
else:
    names = ['NEW', 'pend-weight', 'vertex', 'transparent', 'lobe-select', 'light/BSDF entry', 'shadow-result', 'BSDF-sample']
    stamps = ['camera-ray block', 'traversal', 'vertex/shadow-result block', 'light+BSDF block', 'tail/back-edge']
mode = sys.argv[1] if len(sys.argv) > 1 else 'fast'
which = sys.argv[2] if len(sys.argv) > 2 else 'spheres'
W, H, passes = (int(a) for a in sys.argv[3:6]) if len(sys.argv) > 5 else (1920, 1080, 16)
z = np.load(os.path.join(ROOT, 'tests/golden/scenes.npz'))
a169 = Scene.from_npz(z, 'spheres_a169/', 's')
sc = {'spheres': a169, 'caustics': Scene.from_npz(z, 'caustics_a169/', 'c'), 'stress': stress_scene(a169, 1000, 16)}[which]
with HipRenderer(sc, W, H, counters=True, strict=(mode == 'strict'), exact=(mode == 'exact'), flags=(capi.KAJO_FLAG_DEFERRED if DEFERRED else 0)) as r:
    r.render(passes).wait()
    c = r.counters()
    out = (C.c_ulonglong * 28)()
    capi.check(capi.lib().kajo_hip_debug_profile(r._h, out))
iters = c['laneSlots'] / 64
print('%s %s %dx%d x%d: wave-iterations %.3e, paths %.3e, kernel ms %.2f, trav/path %.3f, vert/path %.3f' % (
    mode, which, W, H, passes, iters, c['paths'], c['kernelMs'], c['traversals'] / c['paths'], c['vertices'] / c['paths']))
for k, n in enumerate(names):
    ex, la = out[2 * k], out[2 * k + 1]
    print('%-18s executed in %5.1f%% of iterations, %4.1f lanes active when executed (%.0f%%)' % (n, 100 * ex / iters, la / max(ex, 1), 100 * la / max(ex, 1) / 64))
st = [out[16 + k] for k in range(9)]
if not DEFERRED and any(st[5:]):  # the large-scene kernels stamp inside their light loop: stamp 3 is then what follows the loop (BSDF sampling)
    # (round 5, the balanced loop: the unit of work is a (vertex, light) pair dealt to any lane whose ray registers are dead)
    stamps = stamps[:3] + ['BSDF sampling (after the light loop)', stamps[4], 'light loop (A): draws of two lights per vertex, which of them count',
                           'light loop (B): pairs dealt, vertices fetched, samples + the queries\' own part', 'light loop: lists walked',
                           'light loop (C): answers back to the vertices\' owners']
tot = sum(st)
for n, v in zip(stamps, st):
    print('%-46s %5.1f%% of wave time, %.0f cycles per iteration' % (n, 100 * v / tot, v / iters))
