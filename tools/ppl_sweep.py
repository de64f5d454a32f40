import sys, os, time, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
from kajo_amd.renderer import HipRenderer
from kajo_amd.scene import Scene
z = np.load('tests/golden/scenes.npz'); sc = Scene.from_npz(z, 'spheres_a169/', 's')
for W, H in ((1920, 1080), (3840, 2160)):
    for mode in ('exact', 'fast'):
        for ppl in (2, 4, 8, 16, 32, 64):
            with HipRenderer(sc, W, H, spp=32, depth_limit=8, exact=mode == 'exact', passes_per_launch=ppl) as r:
                r.render(ppl).wait(); r.render(ppl).wait()
                reps = max(1, 64 // ppl) * (2 if W < 3000 else 1)
                c0 = r.counters()
                for _ in range(reps): r.render(ppl)
                r.wait(); c1 = r.counters()
                ms = (c1['kernelMs'] - c0['kernelMs']) / reps
                print('%dx%d %s ppl %2d: %8.3f ms per launch, %7.1f M paths/s in-kernel' % (W, H, mode, ppl, ms, W * H * 25 * ppl / ms / 1e3), flush=True)
