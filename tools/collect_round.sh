#!/bin/bash
# Build side: copy what tools/profile_round.sh <tag> a|b|c left under gpurun_out/<tag>prof/ (merged back by gpurun) into profiles/ under the
# round's names -- the files the docs cite.   usage: tools/collect_round.sh <tag> [counters]
#   counters: only merge the PMC summaries of stages a and b into profiles/<tag>_counters.json (what stage c's bench line reads)
cd "$(dirname "$0")/.." || exit 1
R=${1:?round tag}; P=gpurun_out/${R}prof
python3 - "$R" <<'PY'
import json, os, sys
R = sys.argv[1]
a = {}
what = {"exact": ("c2", ""), "fast": ("c2", " --fast"), "strict": ("c2", " --strict"), "c4_exact": ("c4", None), "c5_exact": ("c5", None), "c5_fast": ("c5", None)}
for tag, (wl, flag) in what.items():
    path = "gpurun_out/pmc/%s_%s/counters.json" % (R, tag)
    if not os.path.exists(path):
        print("missing", path)
        continue
    for k, d in json.load(open(path)).items():
        d["workload"] = wl
        cmd = ("python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline" + flag) if flag is not None else ("python3 tools/launch_workload.py " + tag.replace("_", " "))
        d["collected"] = ("round %s, MI355X, rocprofv3 --kernel-trace --pmc (six separate passes, tools/pmc.sh / tools/pmc_workload.sh; per counter the median over "
                          "the dispatches of a pass) over " % R) + cmd
        a[k] = d
json.dump(a, open("profiles/%s_counters.json" % R, "w"), indent=1, sort_keys=True)
for k, d in sorted(a.items()):
    print("%-32s %s lane utilisation %.3f, SALU/VALU %.3f, HBM %.1f MB per launch, source %s" % (k, d["workload"], d.get("valu_lane_utilisation", 0), d.get("salu_per_valu", 0),
          d.get("hbm_bytes_per_launch", 0) / 1e6, d.get("kernel_source_hash")))
PY
[ "$2" = counters ] && exit 0
clean() { grep -v amdgpu.ids "$1" > "$2"; }
[ -f $P/bench_final.json ] && cp $P/bench_final.json profiles/${R}_bench.json || cp $P/bench.json profiles/${R}_bench.json
for m in exact fast strict; do
  f=$(ls -t $P/stats_$m/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" profiles/${R}_bench_kernel_stats_$m.csv
  [ -f $P/steady_$m.csv ] && cp $P/steady_$m.csv profiles/${R}_bench_kernel_steady_$m.csv
  [ -f $P/pmc_$m.txt ] && clean $P/pmc_$m.txt profiles/${R}_pmc_$m.txt
done
for t in c4_exact c5_exact c5_fast; do
  [ -f $P/pmc_$t.txt ] && clean $P/pmc_$t.txt profiles/${R}_pmc_$t.txt
  f=$(ls -t gpurun_out/pmc/${R}_$t/stats/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" profiles/${R}_kernel_stats_$t.csv
done
for f in blockprof_exact blockprof_fast blockprof_strict blockprof_exact_caustics blockprof_fast_stress blockprof_exact_stress configs rank_share size_sweep; do
  [ -f $P/$f.txt ] && clean $P/$f.txt profiles/${R}_$f.txt
done
[ -f $P/configs_roofline.json ] && cp $P/configs_roofline.json profiles/${R}_configs_roofline.json
[ -f gpurun_out/${R}_parity_workloads.json ] && cp gpurun_out/${R}_parity_workloads.json profiles/${R}_parity_workloads.json
python3 - "$R" <<'PY'
import json, sys
R = sys.argv[1]
d = json.loads(open("profiles/%s_bench.json" % R).read().strip().splitlines()[-1])
print("%s %.1f M paths/s, %.2f ms/step, kernel %.2f ms, frac %.4f, traffic %s" % (d["config"]["numerics"], d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_per_launch"],
      d["roofline"]["frac"], d["roofline"]["traffic"]))
for m in ("fast_mode", "strict_mode", "exact_mode"):
    if m in d:
        print("  %s %.1f M paths/s, kernel %.2f ms, frac %.4f" % (m, d[m]["value"], d[m]["kernel_ms_per_launch"], d[m]["roofline"]["frac"]))
if d.get("cpu_baseline"):
    print("cpu reference %.1f M paths/s on %d cores, x %.0f" % (d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["speedup_vs_cpu_baseline"]))
PY
for m in exact fast strict; do [ -f profiles/${R}_bench_kernel_steady_$m.csv ] && tail -n +2 profiles/${R}_bench_kernel_steady_$m.csv; done
