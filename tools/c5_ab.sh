#!/bin/bash
# GPU box: configs[4] (1000 spheres / 16 lights, 4K x 32 passes) over the tools' twin and every variant of tools/build_variant.sh
# usage: tools/c5_ab.sh [modes, default fast]
cd "$(dirname "$0")/.." || exit 1 # (the repository root, wherever the script is started from)
mkdir -p gpurun_out
M=${1:-fast}
for rep in 1 2; do
for lib in kajo_amd/libkajo_hip_tune.so kajo_amd/variants/libkajo_hip_*.so; do [ -f $lib ] || continue; echo "== $lib"; KAJO_HIP_LIB=$PWD/$lib python tools/modes.py c5 reps=2 modes=$M 2>>gpurun_out/sweep_errors.log | grep -v "^configs"; done; done
