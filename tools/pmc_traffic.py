"""Aggregate a rocprofv3 --pmc counter_collection CSV by kernel family (development aid for roofline.traffic).
Usage: python tools/pmc_traffic.py <counter_collection.csv> <COUNTER> <iterations>"""
import csv, sys, collections
path, counter, iters = sys.argv[1], sys.argv[2], float(sys.argv[3])
agg = collections.defaultdict(lambda: [0, 0.0])
with open(path) as f:
    for r in csv.DictReader(f):
        if r.get('Counter_Name') != counter:
            continue
        n = r['Kernel_Name']
        fam = ('engine: ' + n.split('<')[0].replace('void ldetr::', '')) if ('gemm_f32' in n or 'gemm_skinny' in n or 'gemm_small' in n or 'gemm_epilogue' in n) else 'other'
        a = agg[fam]; a[0] += 1; a[1] += float(r['Counter_Value'])
tot = sum(v[1] for v in agg.values())
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f'{counter:12s} {k:34s} dispatches/iter={v[0] / iters:8.1f} value/iter={v[1] / iters:14.1f}')
print(f'{counter:12s} TOTAL value/iter={tot / iters:14.1f}')
