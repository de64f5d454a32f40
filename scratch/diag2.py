import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import numpy as np, warnings
warnings.filterwarnings('ignore')
from kajo_amd.renderer import HipRenderer
from kajo_amd.scene import Scene
from oraclelib import OracleLib
import torch
print('torch sees gpu:', torch.cuda.is_available())
z=np.load(os.path.join(ROOT,'tests/golden/scenes.npz'))
sc=Scene.from_npz(z,'spheres_a1/','s')
O=OracleLib('oracle'); ho=O.create(sc,0)
W=H=64
basis=ho.camera_basis()
p1,p2,p3,o=basis.astype(np.float64)
ys,xs=np.mgrid[0:H,0:W]; xs=xs.ravel(); ys=ys.ravel()
sx=(xs+0.5)/W; sy=(H-ys-0.5)/H
d=p1+np.outer(sx,p2-p1)+np.outer(sy,p3-p1)-o; d/=np.linalg.norm(d,axis=1,keepdims=True)
ids=ho.trace(np.repeat(o[None],len(xs),0),d)['idx'].reshape(H,W)
for S,depth,passes in ((32,0,1),(32,1,1),(32,8,4)):
    want=ho.render(W,H,S=S,passes=passes,depth_limit=depth)[...,:3]/passes
    with HipRenderer(sc,W,H,spp=S,depth_limit=depth) as r: got=r.render(passes).radiance()[...,:3]/passes
    ad=np.abs(got-want).max(-1)
    print('S',S,'depth',depth,'median',np.median(ad),'p99',np.percentile(ad,99),'max',ad.max(),'nan',np.isnan(got).sum())
    for i in range(12):
        m=ids==i
        if m.sum(): print('   id',i,'n',m.sum(),'mean abs diff',np.nanmean(ad[m]),'max',np.nanmax(ad[m]),'mean radiance',np.nanmean(want[m]))
