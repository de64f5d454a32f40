#!/bin/bash
# CPU only: kajo_amd/csrc/stage.cpp (object records, uniform grid, per-light visibility lists) under AddressSanitizer + UBSan on the scenes
# of the tests (spheres.json, the dialect scene, 120 / 300 / 1000-sphere scenes, the adversarial geometry of tests/test_shadow_lists_cpu.py).
HERE=$(cd "$(dirname "$0")/.." && pwd); T=$(mktemp -d /tmp/kajo_san.XXXXXX)
python3 - "$T" <<PY || exit 1
import sys; sys.path[:0]=['$HERE','$HERE/tests']
import numpy as np
from kajo_amd.scene import Scene, stress_scene
from test_shadow_lists_cpu import adversarial_scene
T=sys.argv[1]
z=np.load('$HERE/tests/golden/scenes.npz')
a=Scene.from_npz(z,'spheres_a169/','s')
a.write_pod(T+'/spheres.pod'); Scene.from_npz(z,'dialect_a1/','d').write_pod(T+'/dialect.pod')
stress_scene(a,1000,16).write_pod(T+'/stress1000.pod'); stress_scene(a,300,8,seed=77).write_pod(T+'/stress300.pod'); stress_scene(a,120,1,seed=3).write_pod(T+'/stress120.pod')
for s in (1,2,3,4): adversarial_scene(a,s).write_pod(T+'/adv%d.pod'%s)
PY
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -ffp-contract=off -I$HERE/include -I$HERE/kajo_amd/csrc $HERE/tools/stage_san.cpp $HERE/kajo_amd/csrc/stage.cpp -o $T/stage_san || exit 1
$T/stage_san $T/*.pod; rc=$?; rm -rf $T; exit $rc
