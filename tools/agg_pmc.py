"""Per-dispatch averages of rocprofv3 --pmc counters for kernels whose name contains a pattern: python tools/agg_pmc.py <dir or csv> [pattern]"""
import csv, glob, os, sys, collections
root = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else 'gemm_f32_kernel'
paths = [root] if os.path.isfile(root) else glob.glob(root + '/**/*counter_collection.csv', recursive=True)
acc = collections.defaultdict(float); n = collections.Counter()
for path in paths:
    for r in csv.DictReader(open(path)):
        if pat not in r['Kernel_Name']: continue
        key = (r['Kernel_Name'].split('(')[0][-60:], r['Counter_Name'])
        acc[key] += float(r['Counter_Value']); n[key] += 1
for (kn, cn), v in sorted(acc.items()):
    print(f'{kn:60s} {cn:28s} {v / n[(kn, cn)]:16.0f}  x{n[(kn, cn)]}')
