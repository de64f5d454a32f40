#!/bin/bash
# Passes at the end of a launch an idle lane may take over (render_args.h stealWindow; sizes the per-wave mailbox in LDS), configs[1].
cd "$(dirname "$0")/.." || exit 1 # (the repository root, wherever the script is started from)
mkdir -p gpurun_out
for w in 2 3 4 5 6 8; do
  echo "steal window $w:"; KAJO_HIP_LIB=$PWD/kajo_amd/libkajo_hip_tune.so KAJO_STEAL_WINDOW=$w python tools/modes.py c2 reps=6 modes=exact,fast 2>>gpurun_out/sweep_errors.log | grep -E "exact|fast"
done
