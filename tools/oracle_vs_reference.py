#!/usr/bin/env python3
"""The oracle (libm numerics) against the COMPILED REFERENCE (-O2 build, oracle/_ref) over a WHOLE frame, on the CPU: NaN pixels on both
sides, bit-identical share, and the pixels whose relative difference says a path took another decision (none, if the oracle restates the
reference). Needs oracle/_ref (the build container, or a GPU box the built libraries travelled to).
usage: oracle_vs_reference.py <scene key of tests/golden/scenes.npz | mix:<scene seed> (tests/scenes_extra.py)> W H passes     e.g. spheres_a169 1920 1080 16 (2.5 min on 8 cores)"""
import sys, time, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[ROOT,os.path.join(ROOT,'tests')]
import numpy as np
from concurrent.futures import ThreadPoolExecutor
from kajo_amd.scene import Scene
from oraclelib import OracleLib
z=np.load(os.path.join(ROOT,'tests','golden','scenes.npz'))
key, W, H, P = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
if key.startswith('mix:'):
    sys.path.insert(0,os.path.join(ROOT,'tools'))
    from scenes_extra import mixed_scene
    sc=mixed_scene(Scene.from_npz(z,'spheres_a169/','spheres.json 16:9'),int(key[4:]))
else:
    sc=Scene.from_npz(z,key+'/',key)
SEED=0o715517
t=time.time()
orc=OracleLib('oracle').create(sc,0).render(W,H,S=32,passes=P,seed=SEED,depth_limit=8,threads=8)[...,:3]
print('oracle(libm) %.0f s'%(time.time()-t), flush=True)
L=OracleLib('ref_strict')
bands=[(0,y,W,min(8,H-y)) for y in range(0,H,8)]
acc=np.zeros((H,W,4),np.float32)
def job(r):
    q=L.create(sc); a=q.render(W,H,S=32,passes=P,seed=SEED,depth_limit=8,rect=r); q.close(); return r,a
t=time.time()
with ThreadPoolExecutor(8) as ex:
    for (x,y,w,h),a in ex.map(job,bands): acc[y:y+h]=a[y:y+h]
print('reference -O2 %.0f s'%(time.time()-t), flush=True)
ref=acc[...,:3]
both=np.isfinite(orc)&np.isfinite(ref)
d=np.abs(orc-ref); d[~both]=0
rel=d/np.maximum(1.0,np.abs(np.where(both,ref,1.0)))
print(key, W,H,P,'px',W*H,'NaN px oracle',int(np.isnan(orc).any(-1).sum()),'reference',int(np.isnan(ref).any(-1).sum()),'NaN on one side only',int((np.isnan(orc)!=np.isnan(ref)).any(-1).sum()))
print('bit-identical px %.4f'%float(((orc.view(np.uint32)==ref.view(np.uint32))|(np.isnan(orc)&np.isnan(ref))).all(-1).mean()), 'max |d| %.3g'%float(d.max()), 'max rel %.3g'%float(rel.max()), 'px with rel diff > 1e-5:', int((rel.max(-1)>1e-5).sum()), '> 1e-4:', int((rel.max(-1)>1e-4).sum()))
