#!/bin/bash
# What does it cost to keep the waves of a workgroup in step (one barrier per trip)? Round-2 probe: the round-2 kernels built
# -DKAJO_X_LOCKSTEP against the same kernels without it (`make -C kajo_amd/csrc experiments` builds both from the git history).
for lib in kajo_amd/libkajo_hip_r02.so kajo_amd/libkajo_hip_r02_lockstep.so; do
 [ -f "$lib" ] || { echo "$lib missing: make -C kajo_amd/csrc experiments" >&2; exit 1; }
done
for w in 1 2 4; do for lib in kajo_amd/libkajo_hip_r02.so kajo_amd/libkajo_hip_r02_lockstep.so; do
 for mode in "" "--strict"; do
 echo -n "waves/WG $w $lib $mode: "; KAJO_WAVES_PER_BLOCK=$w KAJO_HIP_LIB=$PWD/$lib python bench.py $mode --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f  kernel %.2f ms lane eff %.3f' % (d['value'], d['roofline']['kernel_ms_per_launch'], d['roofline']['lane_efficiency']))"
 done; done; done
