import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import numpy as np, warnings
warnings.filterwarnings('ignore')
from kajo_amd.renderer import HipRenderer
from kajo_amd.scene import Scene
from oraclelib import OracleLib, debug_path
z=np.load(os.path.join(ROOT,'tests/golden/scenes.npz'))
sc=Scene.from_npz(z,'spheres_a1/','s')
O=OracleLib('oracle'); ho=O.create(sc,0)
W=H=128
want=ho.render(W,H,S=1,passes=1,depth_limit=8)[...,:3]
with HipRenderer(sc,W,H,spp=1,depth_limit=8) as r: got=r.render(1).radiance()[...,:3]
ad=np.abs(got-want).max(-1)
rel=ad/np.maximum(np.abs(want).max(-1),1e-3)
bad=np.argwhere(rel>1e-3)
print('paths',W*H,'bad',len(bad))
np.set_printoptions(suppress=True,precision=5,linewidth=200)
for (y,x) in bad[:12]:
    log,rgb=debug_path(ho,W,H,1,int(x),int(y),0)
    print('px',x,y,'got',got[y,x],'want',want[y,x])
    print(log)
