#!/bin/bash
# GPU box: hold thresholds of the large-scene light loop (KAJO_THR_L lanes / KAJO_HOLD_TRIPS trips) on configs[4], a KAJO_TUNING library
# usage: tools/c5_hold_sweep.sh <library> [modes]
cd "$(dirname "$0")/.." || exit 1 # (the repository root, wherever the script is started from)
mkdir -p gpurun_out
LIB=${1:-kajo_amd/libkajo_hip_tune.so}; M=${2:-fast}
for t in "60 6" "56 6" "48 6" "60 3" "48 3" "40 3" "62 8" "60 10" "32 2"; do set -- $t
  echo -n "thrL $1 holdTrips $2: "; KAJO_HIP_LIB=$PWD/$LIB KAJO_THR_L=$1 KAJO_HOLD_TRIPS=$2 python tools/modes.py c5 reps=1 modes=$M 2>>gpurun_out/sweep_errors.log | grep -v "^configs" | tr '\n' ' '; echo
done
