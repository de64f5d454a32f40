#!/bin/bash
# Copy what tools/profile_round.sh left under gpurun_out/<tag>prof/ into profiles/ (the files the docs cite).
R=${1:-r04}; P=gpurun_out/${R}prof
[ -f $P/bench_final.json ] && cp $P/bench_final.json profiles/${R}_bench.json || cp $P/bench.json profiles/${R}_bench.json
grep -v amdgpu.ids $P/configs.txt > profiles/${R}_configs.txt
cp $P/pmc_fast.txt profiles/${R}_pmc_fast.txt; cp $P/pmc_strict.txt profiles/${R}_pmc_strict.txt; cp $P/counters.json profiles/${R}_counters.json
[ -f $P/size_sweep.txt ] && grep -v amdgpu.ids $P/size_sweep.txt > profiles/${R}_size_sweep.txt
for k in fast strict; do f=$(ls -t $P/stats_$k/*/*kernel_stats.csv | head -1); cp $f profiles/${R}_bench_kernel_stats_$k.csv; done
for k in fast strict fast_stress strict_stress; do [ -f $P/blockprof_$k.txt ] && grep -v amdgpu.ids $P/blockprof_$k.txt > profiles/${R}_blockprof_$k.txt; done
[ -f $P/rank_share.txt ] && grep -v amdgpu.ids $P/rank_share.txt > profiles/${R}_rank_share_table.txt
python3 - <<PY
import json
d=json.loads(open('profiles/${R}_bench.json').read().strip().splitlines()[-1])
print('FAST %.1f M paths/s, %.2f ms/step, kernel %.2f ms, frac %.4f, traffic %s, executed %.1f TF'%(d['value'],d['ms_per_step'],d['roofline']['kernel_ms_per_launch'],d['roofline']['frac'],d['roofline']['traffic'],(d['roofline'].get('executed_flops') or {}).get('tflops',0)))
n=d['north_star_mode']; print('north star: %s %.1f, %.0fx, %d/%d px'%(n['numerics'],n['value'],n['speedup_vs_cpu_baseline'],n['bit_identical_px'],n['px']))
print('cpu ref %.1f port %.1f speedup %.0f hash %s'%(d['cpu_baseline']['value'],d['cpu_baseline']['port']['value'],d['speedup_vs_cpu_baseline'],d['config']['kernel_source_hash']))
c=json.load(open('profiles/${R}_counters.json'))
for k,v in c.items(): print(k,'util %.3f'%v['valu_lane_utilisation'],'hbm',v['hbm_bytes_per_launch'],v['kernel_source_hash'],'valu %.3e'%v['counters_per_launch']['SQ_INSTS_VALU'])
PY
head -2 profiles/${R}_bench_kernel_stats_fast.csv | tail -1 | cut -c1-60; head -2 profiles/${R}_bench_kernel_stats_strict.csv | tail -1 | cut -c1-60; cut -c1-100 profiles/${R}_configs.txt
