import sys, os, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT)
import numpy as np, warnings
warnings.filterwarnings('ignore')
from kajo_amd.renderer import HipRenderer
from kajo_amd.scene import Scene, stress_scene
z=np.load(os.path.join(ROOT,'tests/golden/scenes.npz'))
sc=stress_scene(Scene.from_npz(z,'spheres_a169/','s'),1000,16)
for depth in (0,8):
    with HipRenderer(sc,1920,1080,spp=32,depth_limit=depth,counters=True,passes_per_launch=8) as r:
        r.render(8).wait(); c0=r.counters(); t=time.perf_counter(); r.render(8).wait(); dt=time.perf_counter()-t; c1=r.counters()
    tr=c1['traversals']-c0['traversals']; p=c1['paths']-c0['paths']
    print('depth limit %d: %.2f G paths/s, %.2f trav/path, %.2f G traversals/s, lane eff %.3f'%(depth,p/dt/1e9,tr/p,tr/dt/1e9,tr/(c1['laneSlots']-c0['laneSlots'])))
