mkdir -p gpurun_out/r06
python -m pytest tests -m gpu -x -q > gpurun_out/r06/gpu_suite4.log 2>&1; echo "pytest rc $?" >> gpurun_out/r06/gpu_suite4.log; tail -3 gpurun_out/r06/gpu_suite4.log
(for lib in kajo_amd/libkajo_hip_tune.so kajo_amd/variants/libkajo_hip_big5.so; do echo "== c5 $lib"; KAJO_HIP_LIB=$PWD/$lib python tools/modes.py c5 reps=2 modes=exact,fast,strict 2>>gpurun_out/sweep_errors.log | grep -v "^configs"; done
echo "== c5 big5, grid LDS limit 30 KiB"; KAJO_GRID_LDS_LIMIT=30720 KAJO_HIP_LIB=$PWD/kajo_amd/variants/libkajo_hip_big5.so python tools/modes.py c5 reps=2 modes=exact,fast 2>>gpurun_out/sweep_errors.log | grep -v "^configs"
echo "== EXACT hold sweep c2"; bash tools/hold_sweep_exact.sh) > gpurun_out/r06/c5_ab.txt 2>&1
cat gpurun_out/r06/c5_ab.txt
bash tools/pmc_workload.sh r06_c4_exact_b c4 exact > gpurun_out/r06/pmc_c4_exact_b.txt 2>&1; grep "=>\|kajo_render" gpurun_out/r06/pmc_c4_exact_b.txt | tail -8
