#!/bin/bash
# (The stash variant is built from commit 075eefd -- the experiment was removed from the source afterwards: git worktree add /tmp/stash 075eefd, tools/build_variant.sh there.)
# GPU box, round 6: (1) any lane may give a pass away (integrator.inc.hip KAJO_ANY_LANE_GIVES; variant oldgive = rounds 2-5: only lanes between
# two paths), (2) the camera-ray stash experiment (STASH; variant stash: KFLAGS=-DKAJO_STASH=1) -- against the product's kernels (the tools'
# twin). Correctness of the stash variant first (STRICT = oracle bit for bit, EXACT ends every path in the oracle's generator state, any
# cut of the passes one buffer), then in-kernel rates of configs[1], [3], [4] and of short launches, then the block profiles.
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
V=kajo_amd/variants
echo "== correctness, stash variant"
KAJO_HIP_LIB=$PWD/$V/libkajo_hip_stash.so python -m pytest tests/test_hip_parity.py tests/test_hip_exact.py tests/test_hip_pass_cuts.py tests/test_hip_tail_parts.py tests/test_hip_kat.py tests/test_hip_workloads.py -m gpu -q -k "not kajo_render and (not workloads or configs1_whole_frame) and not (one_light_kernel and fast)" 2>&1 | tail -3
run() { KAJO_HIP_LIB=$PWD/$1 python tools/modes.py $2 reps=$4 modes=$3 2>>gpurun_out/sweep_errors.log | grep -v "^configs"; }
for w in c2 c4; do
  for lib in kajo_amd/libkajo_hip_tune.so $V/libkajo_hip_oldgive.so $V/libkajo_hip_stash.so kajo_amd/libkajo_hip_tune.so $V/libkajo_hip_oldgive.so; do
    echo "== $w $lib"; run $lib $w exact,fast,strict 6
  done
done
for lib in kajo_amd/libkajo_hip_tune.so $V/libkajo_hip_oldgive.so; do echo "== c5 $lib"; run $lib c5 exact,fast 2; done
for lib in kajo_amd/libkajo_hip_tune.so $V/libkajo_hip_oldgive.so; do echo "== passes per launch, $lib"; KAJO_HIP_LIB=$PWD/$lib python tools/ppl_sweep.py 2>>gpurun_out/sweep_errors.log | grep 1920; done
echo "== block profile, product kernels (profile twin, 3 waves per SIMD)"
python tools/blockprof.py exact spheres 2>>gpurun_out/sweep_errors.log
echo "== block profile, stash variant (profile twin, 3 waves per SIMD)"
KAJO_HIP_LIB=$PWD/$V/libkajo_hip_stash_prof.so python tools/blockprof.py exact spheres 2>>gpurun_out/sweep_errors.log
