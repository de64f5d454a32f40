# Passes at the end of a launch an idle lane may take over (render_args.h stealWindow; sizes the per-wave mailbox in LDS), configs[1].
cd $GRAFT_REPO_ROOT
for w in 2 3 4 5 6 8; do
  echo "steal window $w:"; KAJO_HIP_LIB=$PWD/kajo_amd/libkajo_hip_tune.so KAJO_STEAL_WINDOW=$w python tools/modes.py c2 reps=6 modes=exact,fast 2>/dev/null | grep -E "exact|fast"
done
