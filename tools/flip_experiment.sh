#!/bin/bash
# flips of the FAST kernels against oracle(libm) under numerics variants (experiment)
cd $GRAFT_REPO_ROOT
run() {
  name="$1"; shift
  touch kajo_amd/csrc/kernel_fast.hip
  make -s -C kajo_amd/csrc "$@" >/dev/null 2>&1 || { echo "$name BUILD FAILED"; return; }
  echo "== $name"
  python tools/parity_outliers.py 16 2>/dev/null | head -8 | grep -E "fast|top   20|without"
  python - <<'PY' 2>/dev/null
import os,sys
sys.path[:0]=['.','tests']
import numpy as np, torch
from kajo_amd.renderer import HipRenderer
from kajo_amd.scene import Scene
from oraclelib import OracleLib
z=np.load('tests/golden/scenes.npz'); scene=Scene.from_npz(z,'spheres_a169/','s')
w,h,p=256,144,16
want=OracleLib('oracle').create(scene,0).render(w,h,S=32,passes=p,seed=0o715517,depth_limit=8,threads=64)[...,:3]/p
r=HipRenderer(scene,w,h,spp=32,depth_limit=8,seed=0o715517); got=r.render(p).radiance()[...,:3]/p; r.close()
d=np.abs(np.clip(got,0,1)-np.clip(want,0,1)).max(-1)
print('   px off by >1e-3: %d, >1e-4: %d, >1e-5: %d, >1e-6: %d of %d'%((d>1e-3).sum(),(d>1e-4).sum(),(d>1e-5).sum(),(d>1e-6).sum(),d.size))
PY
}
run baseline
run nocontract FASTCONTRACT=-ffp-contract=off
run ieee "KFLAGS=-DKAJO_X_IEEE"
run ieee_nocontract FASTCONTRACT=-ffp-contract=off "KFLAGS=-DKAJO_X_IEEE"
run refroots_ieee_nocontract FASTCONTRACT=-ffp-contract=off "KFLAGS=-DKAJO_X_IEEE -DKAJO_X_REFROOTS"
touch kajo_amd/csrc/kernel_fast.hip; make -s -C kajo_amd/csrc >/dev/null 2>&1
