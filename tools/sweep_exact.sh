#!/bin/bash
# GPU box: the STRICT and EXACT builds over (a) the variants of tools/build_variant.sh, (b) hold thresholds of the tools' twin
# library (KAJO_THR_L lanes / KAJO_HOLD_TRIPS trips, integrator.inc.hip MODE_HOLD).   usage: sweep_exact.sh [c2|c4]
cd "$(dirname "$0")/.."
W=${1:-c2}
run() { python tools/modes.py $W reps=4 modes=$1 2>/dev/null | grep -v "^configs" ; }
echo "== product constants, tune twin"; KAJO_HIP_LIB=$PWD/kajo_amd/libkajo_hip_tune.so run strict,exact
for v in kajo_amd/variants/libkajo_hip_*.so; do [ -f $v ] || continue; echo "== $v"; KAJO_HIP_LIB=$PWD/$v run exact; done
echo "== hold sweep (EXACT), tune twin"
for t in "20 1" "28 1" "24 2" "28 2" "32 2" "36 2" "32 3" "40 3" "48 3"; do set -- $t
  echo "thrL $1 holdTrips $2:"; KAJO_HIP_LIB=$PWD/kajo_amd/libkajo_hip_tune.so KAJO_THR_L=$1 KAJO_HOLD_TRIPS=$2 run exact
  if [ $W = c4 ]; then KAJO_HIP_LIB=$PWD/kajo_amd/variants/libkajo_hip_noinl.so KAJO_THR_L=$1 KAJO_HOLD_TRIPS=$2 run exact; fi
done
