"""Summarise rocprofv3 --pmc passes (tools/pmc.sh) per kernel: raw counters per launch, the derived figures bench.py
quotes (HBM bytes with the gfx950 FETCH_SIZE correction, executed FP32 work, VALU lane utilisation), and
<dir>/counters.json in the layout of profiles/r02_counters.json.
usage: pmc_summary.py <dir> [workload-name]"""
import sys, os, csv, glob, collections, json
out = sys.argv[1]
workload = sys.argv[2] if len(sys.argv) > 2 else 'c2'
# Per counter: the MEDIAN over the kernel's dispatches of the pass (the bench's untimed launch with device counters on -- per-wave atomics --
# once showed 198 MB of WRITE_SIZE where every timed launch has 32.4 MB: a mean would carry that into the traffic figure of the timed kernel),
# from the NEWEST run of every pass directory (gpurun merges the files of several calls into one tree on the build side).
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sorted(glob.glob(os.path.join(out, '*', '*'))):
    files = sorted(glob.glob(os.path.join(d, '*counter_collection.csv')), key=os.path.getmtime)
    if not files:
        continue
    for row in csv.DictReader(open(files[-1])):
        vals[row['Kernel_Name']][row['Counter_Name']].append(float(row['Counter_Value']))
agg = {k: {c: sorted(v)[len(v) // 2] if len(v) % 2 else 0.5 * (sorted(v)[len(v) // 2 - 1] + sorted(v)[len(v) // 2]) for c, v in cs.items()} for k, cs in vals.items()}
cnt = {k: {c: len(v) for c, v in cs.items()} for k, cs in vals.items()}
res = {}
for k in sorted(agg):
    if 'kajo_render' not in k: continue
    print('kernel', k)
    per = dict(agg[k])
    for c in sorted(per):
        print('  %-32s per-launch %.6g  (median of %d launches)' % (c, per[c], cnt[k][c]))
    d = {'workload': workload, 'counters_per_launch': per}
    if 'FETCH_SIZE' in per and 'WRITE_SIZE' in per:
        # KiB units; on gfx950 FETCH_SIZE counts 64 of every 128 bytes of a wide coalesced read (MI355X_MICROARCH.md, HBM)
        d['hbm_bytes_per_launch'] = (2 * per['FETCH_SIZE'] + per['WRITE_SIZE']) * 1024
    if 'SQ_THREAD_CYCLES_VALU' in per and 'SQ_ACTIVE_INST_VALU' in per:
        d['valu_lane_utilisation'] = per['SQ_THREAD_CYCLES_VALU'] / (per['SQ_ACTIVE_INST_VALU'] * 64)
    if 'SQ_INSTS_VALU_FLOPS_FP32' in per and 'valu_lane_utilisation' in d:
        # wave-level FLOP count (an FMA counts 2 per lane-slot x 64 lanes) x the share of lanes that were active
        d['executed_fp32_flops_per_launch'] = per['SQ_INSTS_VALU_FLOPS_FP32'] * 64 * d['valu_lane_utilisation']
    if 'SQ_INSTS_SALU' in per and 'SQ_INSTS_VALU' in per:
        d['salu_per_valu'] = per['SQ_INSTS_SALU'] / per['SQ_INSTS_VALU']
    for key in ('hbm_bytes_per_launch', 'valu_lane_utilisation', 'executed_fp32_flops_per_launch', 'salu_per_valu'):
        if key in d: print('  => %-29s %.6g' % (key, d[key]))
    res[k.split('(')[0]] = d
# the kernels these figures belong to: bench.py reports them only while the sources still hash to this value
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
try:
    import importlib.util
    spec = importlib.util.spec_from_file_location('bench_for_hash', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'bench.py'))
    b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
    for d in res.values():
        d['kernel_source_hash'] = b.kernel_source_hash()
except Exception as e:
    print('no source hash:', e)
json.dump(res, open(os.path.join(out, 'counters.json'), 'w'), indent=1, sort_keys=True)
