import csv, sys, collections
path, pat = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(path)):
    if pat in r['Kernel_Name']:
        agg[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
wc = agg.get('SQ_WAVE_CYCLES', 1.0)
for k, v in sorted(agg.items()):
    print(f'{k:28s} per-dispatch={v / max(n[k], 1):16.0f}  /WAVE_CYCLES={v / wc:6.3f}')
