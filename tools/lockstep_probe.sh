for w in 1 2 4; do for lib in kajo_amd/libkajo_hip.so kajo_amd/libkajo_hip_v1.so; do
 for mode in "" "--strict"; do
 echo -n "waves/WG $w $lib $mode: "; KAJO_WAVES_PER_BLOCK=$w KAJO_HIP_LIB=$PWD/$lib python bench.py $mode --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f  kernel %.2f ms lane eff %.3f' % (d['value'], d['roofline']['kernel_ms_per_launch'], d['roofline']['lane_efficiency']))"
 done; done; done
