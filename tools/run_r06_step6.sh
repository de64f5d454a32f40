mkdir -p gpurun_out/r06
(echo "== correctness of the large-scene kernels"; python -m pytest tests/test_hip_parity.py tests/test_hip_ties.py tests/test_hip_open_scenes.py tests/test_hip_edge_cases.py tests/test_hip_fuzz.py tests/test_hip_whole_frames.py tests/test_hip_tail_parts.py tests/test_hip_pass_cuts.py -m gpu -x -q 2>&1 | tail -3
for lib in kajo_amd/variants/libkajo_hip_prev.so kajo_amd/libkajo_hip_tune.so kajo_amd/variants/libkajo_hip_prev.so kajo_amd/libkajo_hip_tune.so; do echo "== c5 $lib"; KAJO_HIP_LIB=$PWD/$lib python tools/modes.py c5 reps=2 modes=exact,fast,strict 2>>gpurun_out/sweep_errors.log | grep -v "^configs"; done
for lib in kajo_amd/variants/libkajo_hip_prev.so kajo_amd/libkajo_hip_tune.so; do echo "== c2 c4 $lib"; KAJO_HIP_LIB=$PWD/$lib python tools/modes.py c2 c4 reps=6 modes=exact,fast 2>>gpurun_out/sweep_errors.log; done) > gpurun_out/r06/c5_ab2.txt 2>&1
cat gpurun_out/r06/c5_ab2.txt
