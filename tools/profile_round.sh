#!/bin/bash
# round-2 profile collection on the GPU box: kernel-trace stats + PMC passes, FAST and STRICT, plus the C5 configs
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02prof
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r02prof/bench.json 2> gpurun_out/r02prof/bench.err
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02prof/stats_fast -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r02prof/stats_fast.log 2>&1
echo "stats fast done"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02prof/stats_strict -- python3 bench.py --strict --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r02prof/stats_strict.log 2>&1
echo "stats strict done"
bash tools/pmc.sh r02_fast > gpurun_out/r02prof/pmc_fast.txt 2>&1
echo "pmc fast done"
bash tools/pmc.sh r02_strict --strict > gpurun_out/r02prof/pmc_strict.txt 2>&1
echo "pmc strict done"
python3 tools/configs.py > gpurun_out/r02prof/configs.txt 2>&1
python3 tools/blockprof.py fast spheres > gpurun_out/r02prof/blockprof_fast.txt 2>&1
python3 tools/blockprof.py strict spheres > gpurun_out/r02prof/blockprof_strict.txt 2>&1
python3 tools/blockprof.py fast stress 1920 1080 8 > gpurun_out/r02prof/blockprof_fast_stress.txt 2>&1
python3 tools/rank_share.py > gpurun_out/r02prof/rank_share.txt 2>&1
echo all done
