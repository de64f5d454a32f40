#!/bin/bash
# GPU box, round 6: the camera-ray stash experiment (integrator.inc.hip STASH; tools/build_variant.sh stash KFLAGS=-DKAJO_STASH=1) and EXACT's
# explicit fused multiply-adds (variant nofma: KFLAGS=-DKAJO_EXACT_FMA=0) against the product's kernels (the tools' twin): correctness of
# the stash variant first (STRICT = oracle bit for bit, EXACT ends every path in the oracle's generator state, any cut of the passes one
# buffer), then in-kernel rates of configs[1] and configs[3], then the block profiles with and without the stash.
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
V=kajo_amd/variants
echo "== correctness, stash variant"
KAJO_HIP_LIB=$PWD/$V/libkajo_hip_stash.so python -m pytest tests/test_hip_parity.py tests/test_hip_exact.py tests/test_hip_pass_cuts.py tests/test_hip_tail_parts.py tests/test_hip_kat.py tests/test_hip_workloads.py -m gpu -x -q -k "not kajo_render and (not workloads or configs1_whole_frame)" 2>&1 | tail -3
run() { KAJO_HIP_LIB=$PWD/$1 python tools/modes.py $2 reps=6 modes=$3 2>>gpurun_out/sweep_errors.log | grep -v "^configs"; }
for w in c2 c4; do
  for lib in kajo_amd/libkajo_hip_tune.so $V/libkajo_hip_stash.so $V/libkajo_hip_nofma.so kajo_amd/libkajo_hip_tune.so $V/libkajo_hip_stash.so; do
    echo "== $w $lib"; run $lib $w exact,fast,strict
  done
done
echo "== block profile, product kernels (profile twin, 3 waves per SIMD)"
python tools/blockprof.py exact spheres 2>>gpurun_out/sweep_errors.log
python tools/blockprof.py fast spheres 2>>gpurun_out/sweep_errors.log
echo "== block profile, stash variant (profile twin, 3 waves per SIMD)"
KAJO_HIP_LIB=$PWD/$V/libkajo_hip_stash_prof.so python tools/blockprof.py exact spheres 2>>gpurun_out/sweep_errors.log
KAJO_HIP_LIB=$PWD/$V/libkajo_hip_stash_prof.so python tools/blockprof.py fast spheres 2>>gpurun_out/sweep_errors.log
