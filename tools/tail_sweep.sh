#!/bin/bash
# The launch tail (capi.cpp partTheTail): how many of the cheapest blocks to render in four parts, in eighths of the chip's wave slots.
# In-kernel rates of configs[1], EXACT and FAST; 0 = no parts.
cd "$(dirname "$0")/.." || exit 1 # (the repository root, wherever the script is started from)
mkdir -p gpurun_out
for q in 0 2 3 4 5 6 8 12; do
  echo "q4 $q:"; KAJO_HIP_LIB=$PWD/kajo_amd/libkajo_hip_tune.so KAJO_TAIL_Q4=$q python tools/modes.py c2 ${CASES:-} reps=6 modes=exact,fast 2>>gpurun_out/sweep_errors.log | grep -E "exact|fast"
done
