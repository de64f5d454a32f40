#!/bin/bash
cd "$(dirname "$0")/.." || exit 1 # (the repository root, wherever the script is started from)
mkdir -p gpurun_out
for t in "36 2" "28 2" "24 2" "20 2" "32 1" "24 1" "16 1" "28 3"; do set -- $t
  echo -n "thrL $1 holdTrips $2: "; KAJO_HIP_LIB=$PWD/kajo_amd/libkajo_hip_tune.so KAJO_THR_L=$1 KAJO_HOLD_TRIPS=$2 python tools/modes.py c2 reps=4 modes=exact 2>>gpurun_out/sweep_errors.log | grep exact
done
