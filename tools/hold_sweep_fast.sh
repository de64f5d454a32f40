#!/bin/bash
# Hold policy of the FAST one-light instance (integrator.inc.hip MODE_HOLD: lanes that must want the light / BSDF blocks, trips they may be put off), configs[1]
cd "$(dirname "$0")/.." || exit 1 # (the repository root, wherever the script is started from)
mkdir -p gpurun_out
for t in "20 1" "12 1" "16 1" "24 1" "28 1" "16 2" "20 2" "24 2" "28 2" "32 2" "36 2" "32 3"; do set -- $t
  echo -n "thrL $1 holdTrips $2: "; KAJO_HIP_LIB=$PWD/kajo_amd/libkajo_hip_tune.so KAJO_THR_L=$1 KAJO_HOLD_TRIPS=$2 python tools/modes.py c2 reps=6 modes=fast 2>>gpurun_out/sweep_errors.log | grep fast
done
