cd $GRAFT_REPO_ROOT
export KAJO_HIP_LIB=$PWD/kajo_amd/libkajo_hip_tune.so
for t in "0 0" "2 2"; do set -- $t
  echo "q4 $1 q2 $2:"
  KAJO_TAIL_Q4=$1 KAJO_TAIL_Q2=$2 python tools/modes.py c4 reps=4 modes=exact,fast 2>/dev/null | grep -E "exact|fast"
  KAJO_TAIL_Q4=$1 KAJO_TAIL_Q2=$2 python tools/configs.py exact "C3/GPU" "C5 at 1080p 8x" 2>/dev/null
  KAJO_TAIL_Q4=$1 KAJO_TAIL_Q2=$2 python tools/configs.py "C3/GPU" "C5 at 1080p 8x" "C5 4K 32" 2>/dev/null
done
