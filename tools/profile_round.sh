#!/bin/bash
# The whole evidence run of a round in one `gpurun` call: bench line, `rocprofv3 --kernel-trace --stats` (FAST, STRICT), the PMC
# passes (counters only ever with --kernel-trace, separate passes), the other BASELINE configs, block profiles.
#   tools/profile_round.sh [round tag, default r05]
# Results under gpurun_out/<tag>prof/; copy what is to be judged into profiles/ (bench.py reads profiles/<tag>_counters.json,
# which carries the hash of the kernel sources it was collected on: stale counters are not reported).
# Needs the diagnostic twins: make -C kajo_amd/csrc prof   (built here: the GPU box has the same toolchain)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=${1:-r05}
OUT=gpurun_out/${R}prof
mkdir -p $OUT
[ -f kajo_amd/libkajo_hip_prof.so ] || make -s -C kajo_amd/csrc prof || { echo "no profile twin" >&2; exit 1; }
[ -f kajo_amd/libkajo_hip_count.so ] || make -s -C kajo_amd/csrc count || echo "no counting twin: the grid scene gets no roofline line" >&2
python3 bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_exact -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --sustain-seconds 0 > $OUT/stats_exact.log 2>&1
echo "stats exact done (the build bench.py times by default)"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_fast -- python3 bench.py --fast --steps 10 --warmup 2 --no-cpu-baseline --sustain-seconds 0 > $OUT/stats_fast.log 2>&1
echo "stats fast done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_strict -- python3 bench.py --strict --steps 5 --warmup 2 --no-cpu-baseline --sustain-seconds 0 > $OUT/stats_strict.log 2>&1
echo "stats strict done"
bash tools/pmc.sh ${R}_exact > $OUT/pmc_exact.txt 2>&1
echo "pmc exact done"
bash tools/pmc.sh ${R}_fast --fast > $OUT/pmc_fast.txt 2>&1
echo "pmc fast done"
bash tools/pmc.sh ${R}_strict --strict > $OUT/pmc_strict.txt 2>&1
echo "pmc strict done"
python3 - <<PY
import json
a = {}
for m in ("exact", "fast", "strict"):
    a.update(json.load(open("gpurun_out/pmc/${R}_%s/counters.json" % m)))
for k, d in a.items():
    d["collected"] = "round ${R}, MI355X, rocprofv3 --kernel-trace --pmc (six separate passes, tools/pmc.sh; per counter the median over the dispatches of a pass) over python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline" + (" --strict" if "strict" in k else " --fast" if "fast" in k else "")
json.dump(a, open("$OUT/counters.json", "w"), indent=1, sort_keys=True)
PY
(echo "== EXACT"; python3 tools/configs.py exact; echo "== FAST"; python3 tools/configs.py; echo "== STRICT"; python3 tools/configs.py strict) > $OUT/configs.txt 2>&1
python3 tools/configs_roofline.py $OUT/configs_roofline.json > $OUT/configs_roofline.txt 2>&1
echo "configs done"
python3 tools/blockprof.py exact spheres > $OUT/blockprof_exact.txt 2>&1
python3 tools/blockprof.py fast spheres > $OUT/blockprof_fast.txt 2>&1
python3 tools/blockprof.py strict spheres > $OUT/blockprof_strict.txt 2>&1
python3 tools/blockprof.py fast stress 1920 1080 8 > $OUT/blockprof_fast_stress.txt 2>&1
echo profiles done
# the bench line again, now that <tag>_counters.json of these kernels exists (roofline.traffic / executed_flops are read from it)
cp $OUT/counters.json profiles/${R}_counters.json
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_final.json 2> $OUT/bench_final.err
python3 tools/rank_share.py > $OUT/rank_share.txt 2>&1
python3 tools/size_sweep.py > $OUT/size_sweep.txt 2>&1
python3 tools/blockprof.py strict stress 1920 1080 4 > $OUT/blockprof_strict_stress.txt 2>&1
echo final bench done
