import sys, os, time
ROOT='/root/repo' if os.path.exists('/root/repo/kajo_amd') else os.getcwd()
sys.path.insert(0,ROOT)
import numpy as np, warnings
warnings.filterwarnings('ignore')
from kajo_amd.renderer import HipRenderer
from kajo_amd.scene import Scene
z=np.load(os.path.join(ROOT,'tests/golden/scenes.npz'))
sc=Scene.from_npz(z,'spheres_a169/','s')
for W,H in ((1920,1080),(3840,2160),(7680,4320)):
  for ppl in (16,64):
    with HipRenderer(sc,W,H,passes_per_launch=ppl) as r:
        r.render(ppl).wait(); c0=r.counters(); r.render(ppl).wait(); c1=r.counters()
    ms=c1['kernelMs']-c0['kernelMs']; p=c1['paths']-c0['paths']
    print('%dx%d ppl %d: kernel %.2f ms, %.1f G paths/s in-kernel'%(W,H,ppl,ms,p/ms/1e6),flush=True)
