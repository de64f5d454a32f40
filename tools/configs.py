"""Throughput of the BASELINE.json configs other than the bench's (parity-test cases, not bench lines)."""
import sys, os, json, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import numpy as np, warnings
warnings.filterwarnings('ignore')
from kajo_amd.renderer import HipRenderer
from kajo_amd.scene import Scene, caustics_scene, stress_scene
z=np.load(os.path.join(ROOT,'tests/golden/scenes.npz'))
a1=Scene.from_npz(z,'spheres_a1/','spheres a1'); a169=Scene.from_npz(z,'spheres_a169/','spheres 16:9')
caus=Scene.from_npz(z,'caustics_a169/','caustics (3 lights)')
cases=[('C1 256x256 S=16 depth1 1 pass',a1,256,256,16,1,1),
       ('C2 1080p 16xS32',a169,1920,1080,32,16,8),
       ('C2 alt: 1080p 1 pass x S=512 (n=22, 484 paths/px)',a169,1920,1080,512,1,8),
       ('C3/GPU: 4K 16xS32 (1 GPU, 512 of 2048 spp)',a169,3840,2160,32,16,8),
       ('C4 caustics 1080p 16xS32 (of 128) depth 8',caus,1920,1080,32,16,8),
       ('C5 1000 spheres/16 lights 4K 1xS32 (of 32)',stress_scene(a169,1000,16),3840,2160,32,1,8),
       ('C5 at 1080p 1xS32',stress_scene(a169,1000,16),1920,1080,32,1,8),
       ('C5 at 1080p 8xS32',stress_scene(a169,1000,16),1920,1080,32,8,8),
       ('C5 4K 32xS32 (all of its 1024 spp, 32 passes per launch)',stress_scene(a169,1000,16),3840,2160,32,32,8)]
sel=sys.argv[1:] 
strict = 'strict' in sel
exact = 'exact' in sel
nolists = 'nolists' in sel   # large scenes: shadow rays through the grid (round 3's schedule) instead of the lights' visibility lists
sel = [x for x in sel if x not in ('strict', 'exact', 'nolists')]
FLAGS = 128 if nolists else 0
for name,sc,W,H,S,passes,depth in cases:
    if sel and not any(s in name for s in sel): continue
    # rate: a handle WITHOUT device counters (their per-wave atomics weigh on launches of many short waves), several launches back to back
    with HipRenderer(sc,W,H,spp=S,depth_limit=depth,strict=strict,exact=exact,passes_per_launch=passes,flags=FLAGS) as r:
        r.render(passes).wait(); r.render(passes).wait()          # warm (the first launch also records the launch order)
        reps = 20 if W*H*passes < 4e6 else (3 if W*H*passes < 2e8 else 1)
        c0=r.counters(); t=time.perf_counter()
        for _ in range(reps): r.render(passes)
        r.wait(); dt=(time.perf_counter()-t)/reps; c1=r.counters()
    ms=(c1['kernelMs']-c0['kernelMs'])/reps
    # work per path: a handle with counters
    with HipRenderer(sc,W,H,spp=S,depth_limit=depth,strict=strict,exact=exact,counters=True,passes_per_launch=passes,flags=FLAGS) as r:
        c=r.render(passes).counters()
    paths=c['paths']
    print('%-48s %8.1f Mpaths/s wall, kernel %8.3f ms, %5.2f trav/path (+ %4.2f shadow queries from light lists), %5.2f vert/path, lane eff %.3f'%(
        name, paths/dt/1e6, ms, c['traversals']/paths, c.get('shadowQueries',0)/paths, c['vertices']/paths, c['traversals']/max(1,c['laneSlots'])), flush=True)
