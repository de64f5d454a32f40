#!/bin/bash
cd "$(dirname "$0")/.." || exit 1 # (the repository root, wherever the script is started from)
mkdir -p gpurun_out
export KAJO_HIP_LIB=$PWD/kajo_amd/libkajo_hip_tune.so
for c in 0.5 1.0 1.5 2.0 3.0; do echo -n "cells/sphere $c: "; KAJO_GRID_CELLS_PER_SPHERE=$c python tools/modes.py c5 reps=1 modes=fast 2>>gpurun_out/sweep_errors.log | grep fast; done
for b in 48 64 96 128; do echo -n "bins $b: "; KAJO_SHADOW_BINS=$b python tools/modes.py c5 reps=1 modes=fast 2>>gpurun_out/sweep_errors.log | grep fast; done
