"""Steady-state launch statistics from a `rocprofv3 --kernel-trace` run: per render kernel the durations of its dispatches WITHOUT the
first two of the process (a handle's first launch runs in image order and measures the blocks, its second is the first in cost order
and, with it, the first of the parted tail; `--stats` averages them in) -- count, mean, median, min, max in ms -- next to the all-
dispatch mean `--stats` reports. usage: steady_stats.py <rocprofv3 output dir> [skip=2]"""
import csv, glob, os, sys, collections
d = sys.argv[1]
skip = next((int(a[5:]) for a in sys.argv[2:] if a.startswith("skip=")), 2)
files = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
if not files:
    sys.exit("no kernel trace under " + d)
rows = collections.defaultdict(list)
for r in csv.DictReader(open(files[-1])):
    if "kajo_render" in r["Kernel_Name"]:
        rows[r["Kernel_Name"].split("(")[0]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
print("kernel,dispatches,mean_ms_all_dispatches,steady_dispatches,steady_mean_ms,steady_median_ms,steady_min_ms,steady_max_ms")
for k, v in sorted(rows.items()):
    v.sort()
    ms = [(e - s) * 1e-6 for s, e in v]
    st = sorted(ms[skip:]) or sorted(ms)
    med = st[len(st) // 2] if len(st) % 2 else 0.5 * (st[len(st) // 2 - 1] + st[len(st) // 2])
    print("%s,%d,%.4f,%d,%.4f,%.4f,%.4f,%.4f" % (k, len(ms), sum(ms) / len(ms), len(st), sum(st) / len(st), med, st[0], st[-1]))
