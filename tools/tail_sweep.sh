# The launch tail (capi.cpp partTheTail): how many of the cheapest blocks to render in 4 / 2 parts, in eighths of the chip's wave slots.
# In-kernel rates of configs[1] (and the caustics scene), EXACT and FAST; "0 0" = no parts.
cd $GRAFT_REPO_ROOT
for t in "0 0" "4 4" "2 2" "2 4" "4 2" "3 3" "6 4" "4 6" "6 6" "2 6" "1 3" "4 3" "3 5"; do set -- $t
  echo "q4 $1 q2 $2:"; KAJO_HIP_LIB=$PWD/kajo_amd/libkajo_hip_tune.so KAJO_TAIL_Q4=$1 KAJO_TAIL_Q2=$2 python tools/modes.py c2 ${CASES:-} reps=6 modes=exact,fast 2>/dev/null | grep -E "exact|fast"
done
