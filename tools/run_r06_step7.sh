mkdir -p gpurun_out/r06
(for lib in kajo_amd/variants/libkajo_hip_neither.so kajo_amd/variants/libkajo_hip_chunk.so kajo_amd/variants/libkajo_hip_gridpf.so kajo_amd/libkajo_hip_tune.so; do echo "== c5 $lib"; KAJO_HIP_LIB=$PWD/$lib python tools/modes.py c5 reps=2 modes=exact,fast,strict 2>>gpurun_out/sweep_errors.log | grep -v "^configs"; done) > gpurun_out/r06/c5_ab3.txt 2>&1
cat gpurun_out/r06/c5_ab3.txt
