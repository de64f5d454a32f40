import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import numpy as np, warnings
warnings.filterwarnings('ignore')
from kajo_amd.renderer import HipRenderer
from kajo_amd.scene import Scene
from oraclelib import OracleLib
z=np.load(os.path.join(ROOT,'tests/golden/scenes.npz'))
sc=Scene.from_npz(z,'spheres_a1/','s')
O=OracleLib('oracle'); ho=O.create(sc,1)
W=H=64
basis=ho.camera_basis()
for S,depth in ((1,0),(1,1),(1,8),(16,1)):
    want=ho.render(W,H,S=S,passes=1,depth_limit=depth)
    with HipRenderer(sc,W,H,spp=S,depth_limit=depth,strict=True) as r: got=r.render(1).radiance()
    diff=(got[...,:3].view(np.uint32)!=want[...,:3].view(np.uint32)).any(-1)
    print('S',S,'depth',depth,'differing pixels',diff.sum(),'of',W*H, 'maxabs', np.abs(got[...,:3]-want[...,:3]).max())
    if diff.sum() and S==1:
        ys,xs=np.nonzero(diff)
        # primary hit of pixel centers (approx) via oracle trace
        p1,p2,p3,o=basis.astype(np.float64)
        sx=(xs+0.5)/W; sy=(H-ys+0.5)/H
        d=p1+np.outer(sx,p2-p1)+np.outer(sy,p3-p1)-o; d/=np.linalg.norm(d,axis=1,keepdims=True)
        tr=ho.trace(np.repeat(o[None],len(xs),0),d)
        print(' primary ids of differing px:', np.bincount(tr['idx'],minlength=12))
        for k in range(min(6,len(xs))):
            print('  px',xs[k],ys[k],'id',tr['idx'][k],'got',got[ys[k],xs[k],:3],'want',want[ys[k],xs[k],:3])
