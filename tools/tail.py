"""How much of a launch is tail (last workgroups draining)? ms per pass as a function of passes per launch."""
import sys, os, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT)
import numpy as np
from kajo_amd.renderer import HipRenderer
from kajo_amd.scene import Scene
z=np.load(os.path.join(ROOT,'tests/golden/scenes.npz'))
sc=Scene.from_npz(z,'spheres_a169/','s')
for ppl in (1,2,4,8,16,32,64,128):
    with HipRenderer(sc,1920,1080,passes_per_launch=ppl) as r:
        r.render(ppl).wait(); c0=r.counters()
        n=max(1,128//ppl)
        for _ in range(n): r.render(ppl)
        r.wait(); c1=r.counters()
    ms=(c1['kernelMs']-c0['kernelMs'])/(n*ppl)
    print('passes/launch %3d: %.3f ms per pass (%.1f Gpaths/s in-kernel)'%(ppl,ms,1920*1080*25/ms/1e6))
