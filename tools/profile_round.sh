#!/bin/bash
# The evidence run of a round, in three `gpurun` calls (a call is limited to 20 minutes): bench line, `rocprofv3 --kernel-trace --stats`
# (EXACT, FAST, STRICT: long enough runs that a handle's cold launches do not carry the average, plus tools/steady_stats.py: the same
# trace without each process's first two dispatches), the PMC passes (counters only ever with --kernel-trace, separate passes) of the
# headline kernels AND of the kernels configs[3] / configs[4] run, the other BASELINE configs, block profiles, rank shares.
#   tools/profile_round.sh <round tag, e.g. r06> a|b|c
#     a  bench, stats x 3, PMC of kajo_render_{exact,fast,strict} on configs[1]
#     b  stats + PMC of kajo_render_exact_lights (configs[3]) and kajo_render_{exact,fast}_biglist* (configs[4] at 4K x 32)
#     c  (after the counters of a + b are merged into profiles/<tag>_counters.json on the build side: tools/collect_round.sh <tag> counters)
#        the bench line again -- it reads roofline.traffic / executed_flops from that file --, configs, roofline per config, block profiles,
#        rank shares, size sweep
# Results under gpurun_out/<tag>prof/; tools/collect_round.sh <tag> copies what is to be judged into profiles/.
# Needs the diagnostic twins (make -C kajo_amd/csrc prof count), built on the build side so that they travel.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
R=${1:?round tag}; STAGE=${2:?stage a, b or c}
OUT=gpurun_out/${R}prof
mkdir -p $OUT
case $STAGE in
a)
  python3 bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench done"
  for m in exact fast strict; do
    flag=$([ $m = exact ] && echo "" || echo "--$m")
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$m -- python3 bench.py $flag --steps 40 --warmup 5 --no-cpu-baseline --sustain-seconds 0 > $OUT/stats_$m.log 2>&1
    python3 tools/steady_stats.py $OUT/stats_$m > $OUT/steady_$m.csv 2>&1
    echo "stats $m done"; cat $OUT/steady_$m.csv
  done
  bash tools/pmc.sh ${R}_exact > $OUT/pmc_exact.txt 2>&1; echo "pmc exact done"
  bash tools/pmc.sh ${R}_fast --fast > $OUT/pmc_fast.txt 2>&1; echo "pmc fast done"
  bash tools/pmc.sh ${R}_strict --strict > $OUT/pmc_strict.txt 2>&1; echo "pmc strict done"
  ;;
b)
  bash tools/pmc_workload.sh ${R}_c4_exact c4 exact > $OUT/pmc_c4_exact.txt 2>&1; echo "c4 exact done"
  bash tools/pmc_workload.sh ${R}_c5_exact c5 exact > $OUT/pmc_c5_exact.txt 2>&1; echo "c5 exact done"
  bash tools/pmc_workload.sh ${R}_c5_fast c5 fast > $OUT/pmc_c5_fast.txt 2>&1; echo "c5 fast done"
  python3 tools/blockprof.py exact spheres > $OUT/blockprof_exact.txt 2>&1
  python3 tools/blockprof.py fast spheres > $OUT/blockprof_fast.txt 2>&1
  python3 tools/blockprof.py strict spheres > $OUT/blockprof_strict.txt 2>&1
  python3 tools/blockprof.py exact caustics > $OUT/blockprof_exact_caustics.txt 2>&1
  python3 tools/blockprof.py fast stress 1920 1080 8 > $OUT/blockprof_fast_stress.txt 2>&1
  python3 tools/blockprof.py exact stress 1920 1080 8 > $OUT/blockprof_exact_stress.txt 2>&1
  echo "block profiles done"
  ;;
c)
  python3 bench.py --steps 20 --warmup 5 > $OUT/bench_final.json 2> $OUT/bench_final.err; echo "final bench done"
  (echo "== EXACT"; python3 tools/configs.py exact; echo "== FAST"; python3 tools/configs.py; echo "== STRICT"; python3 tools/configs.py strict) > $OUT/configs.txt 2>&1
  python3 tools/configs_roofline.py $OUT/configs_roofline.json > $OUT/configs_roofline.txt 2>&1
  echo "configs done"
  python3 tools/rank_share.py > $OUT/rank_share.txt 2>&1
  python3 tools/size_sweep.py > $OUT/size_sweep.txt 2>&1
  echo "rank shares, size sweep done"
  ;;
esac
