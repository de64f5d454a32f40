# Hold policy of the FAST one-light instance (integrator.inc.hip MODE_HOLD: lanes that must want the light / BSDF blocks, trips they may be put off), configs[1]
cd $GRAFT_REPO_ROOT
for t in "20 1" "12 1" "16 1" "24 1" "28 1" "16 2" "20 2" "24 2" "28 2" "32 2" "36 2" "32 3"; do set -- $t
  echo -n "thrL $1 holdTrips $2: "; KAJO_HIP_LIB=$PWD/kajo_amd/libkajo_hip_tune.so KAJO_THR_L=$1 KAJO_HOLD_TRIPS=$2 python tools/modes.py c2 reps=6 modes=fast 2>/dev/null | grep fast
done
