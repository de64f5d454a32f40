import os, sys
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[ROOT, os.path.join(ROOT,'tests')]
import numpy as np, warnings
warnings.filterwarnings('ignore')
from kajo_amd import capi
from kajo_amd.renderer import HipRenderer
from kajo_amd.scene import Scene
from oraclelib import OracleLib
z=np.load(os.path.join(ROOT,'tests/golden/scenes.npz'))
O=OracleLib('oracle'); F=capi.KAJO_FLAG_DEFERRED
sc=Scene.from_npz(z,'spheres_a169/','s')
W,H,S,passes,depth=160,90,32,4,8
want=O.create(sc,math=1).render(W,H,S=S,passes=passes,seed=0o715517,depth_limit=depth)
def cmp(tag,**kw):
    with HipRenderer(sc,W,H,spp=S,seed=0o715517,depth_limit=depth,strict=True,flags=F,**kw) as r:
        got=r.render(passes).radiance()
    a,b=got[...,:3],want[...,:3]
    bad=~((a.view(np.uint32)==b.view(np.uint32))|(np.isnan(a)&np.isnan(b))).all(axis=-1)
    print(tag, 'bad px', int(bad.sum()), 'first', np.argwhere(bad)[:5].tolist(), flush=True)
for knobs in ({}, {'KAJO_STEAL_WINDOW':'4'}, {'KAJO_STASH_DEPTH':'2'}, {'KAJO_RING_SLOTS':'4'}, {'KAJO_STASH_DEPTH':'2','KAJO_RING_SLOTS':'4','KAJO_THR_L':'40','KAJO_THR_STALL':'12'}):
    for k in ('KAJO_STASH_DEPTH','KAJO_RING_SLOTS','KAJO_THR_L','KAJO_THR_STALL','KAJO_STEAL_WINDOW'): os.environ.pop(k,None)
    os.environ.update(knobs)
    for ppl in (2,16):
        cmp('%s ppl=%d'%(knobs,ppl), passes_per_launch=ppl)
