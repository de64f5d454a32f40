"""STRICT kernels vs oracle(strict) on one frame: which pixels differ, by how much, with / without wave splitting."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, warnings
warnings.filterwarnings('ignore')
from kajo_amd.renderer import HipRenderer
from kajo_amd.scene import Scene
from kajo_amd import capi
from oraclelib import OracleLib
W, H, P = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (256, 144, 16)
z = np.load(os.path.join(ROOT, 'tests/golden/scenes.npz'))
sc = Scene.from_npz(z, 'spheres_a169/', 's')
want = OracleLib('oracle').create(sc, 1).render(W, H, S=32, passes=P, seed=0o715517, depth_limit=8, threads=16)
for name, flags, ppl in (('default', 0, 0), ('default again', 0, 0), ('no split', capi.KAJO_FLAG_NO_SPLIT, 0), ('ppl 4', 0, 4), ('no split no reorder', capi.KAJO_FLAG_NO_SPLIT | capi.KAJO_FLAG_NO_REORDER, 0)):
    with HipRenderer(sc, W, H, spp=32, depth_limit=8, seed=0o715517, strict=True, flags=flags, passes_per_launch=ppl) as r:
        got = r.render(P).radiance()
    a, b = got[..., :3], want[..., :3]
    d = (a.view(np.uint32) != b.view(np.uint32)) & ~(np.isnan(a) & np.isnan(b))
    px = np.argwhere(d.any(-1))
    print(name, 'differing px', len(px))
    for y, x in px[:6]:
        print('   ', (x, y), a[y, x], b[y, x], a[y, x].view(np.uint32), b[y, x].view(np.uint32))
