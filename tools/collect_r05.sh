#!/bin/bash
# Copy what the last `tools/profile_round.sh r05` call merged back (gpurun_out/r05prof/) into profiles/ under the round's names.
cd "$(dirname "$0")/.."
O=gpurun_out/r05prof
cp $O/bench_final.json profiles/r05_bench.json
cp $O/counters.json profiles/r05_counters.json
for m in exact fast strict; do cp "$(ls -t $O/stats_$m/runc/*_kernel_stats.csv | head -1)" profiles/r05_bench_kernel_stats_$m.csv; cp $O/pmc_$m.txt profiles/r05_pmc_$m.txt; done
for f in blockprof_exact blockprof_fast blockprof_strict blockprof_fast_stress blockprof_strict_stress configs rank_share size_sweep; do cp $O/$f.txt profiles/r05_$f.txt; done
cp $O/configs_roofline.json profiles/r05_configs_roofline.json
[ -f gpurun_out/r05_parity_workloads.json ] && cp gpurun_out/r05_parity_workloads.json profiles/r05_parity_workloads.json
[ -f gpurun_out/pmc/r05_c5_fast/summary.txt ] && cp gpurun_out/pmc/r05_c5_fast/summary.txt profiles/r05_pmc_c5_traffic.txt
echo collected
