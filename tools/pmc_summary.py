import sys, os, csv, glob, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(os.path.join(out, '*', '*', '*counter_collection.csv')):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name']
        agg[k][row['Counter_Name']] += float(row['Counter_Value'])
        cnt[k][row['Counter_Name']] += 1
for k in agg:
    if 'kajo_render' not in k: continue
    print('kernel', k)
    for c in sorted(agg[k]):
        print('  %-28s per-launch %.6g  (launches %d)' % (c, agg[k][c] / cnt[k][c], cnt[k][c]))
