"""One rank's share of BASELINE configs[2] (3840x2160, 64 passes) on ONE GPU: tiles dealt as if N ranks rendered the
frame, rank 0's tiles rendered here. Rate per rank x N = what N GPUs deliver apart from the gather (strong scaling).
usage: rank_share.py [fast|strict|exact]   (default exact: the build bench.py times at every N)"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, warnings
warnings.filterwarnings('ignore')
from kajo_amd.renderer import HipRenderer
from kajo_amd.scene import Scene
z = np.load(os.path.join(ROOT, 'tests/golden/scenes.npz'))
sc = Scene.from_npz(z, 'spheres_a169/', 's')
W, H, P = 3840, 2160, 64
MODE = (sys.argv[1:] + ['exact'])[0]
print('numerics', MODE, flush=True)
for N in (1, 2, 4, 8):
    for ppl in (64,):
        with HipRenderer(sc, W, H, tile_index=0, tile_count=N, passes_per_launch=ppl, strict=(MODE == 'strict'), exact=(MODE == 'exact')) as r:
            r.render(P).wait()
            c0 = r.counters(); t = time.perf_counter(); r.render(P).wait(); dt = time.perf_counter() - t; c1 = r.counters()
        paths = c1['paths'] - c0['paths']
        print('N=%d ppl=%2d: rank share %.1f ms, %.0f Mpaths/s per rank, x N = %.0f, vs N=1 efficiency see ratio' % (N, ppl, dt * 1e3, paths / dt / 1e6, N * paths / dt / 1e6), flush=True)
