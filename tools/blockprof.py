import sys, os, ctypes as C
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT)
import numpy as np, warnings
warnings.filterwarnings('ignore')
from kajo_amd.renderer import HipRenderer
from kajo_amd.scene import Scene
from kajo_amd import capi
z=np.load(os.path.join(ROOT,'tests/golden/scenes.npz'))
sc=Scene.from_npz(z,'spheres_a169/','s')
with HipRenderer(sc,1920,1080,counters=True) as r:
    r.render(16).wait()
    c=r.counters()
    out=(C.c_ulonglong*28)()
    capi.check(capi.lib().kajo_hip_debug_profile(r._h, out))
iters=c['laneSlots']/64
names=['NEW','pend-weight','vertex','transparent','lobe-select','light/BSDF entry','shadow-result','BSDF-sample']
print('wave-iterations %.3e, paths %.3e, kernel ms %.2f'%(iters,c['paths'],c['kernelMs']))
for k,n in enumerate(names):
    ex,la=out[2*k],out[2*k+1]
    print('%-18s executed in %5.1f%% of iterations, %4.1f lanes active when executed (%.0f%%)'%(n,100*ex/iters,la/max(ex,1),100*la/max(ex,1)/64))

st=[out[16+k] for k in range(5)]
tot=sum(st)
for n,v in zip(['camera-ray block','traversal','vertex/shadow-result block','light+BSDF block','tail/back-edge'],st):
    print('%-28s %5.1f%% of wave time, %.0f cycles per iteration'%(n,100*v/tot,v/iters))
