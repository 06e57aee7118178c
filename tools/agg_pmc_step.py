"""Per-kernel totals of the counters of one `rocprofv3 --pmc ... -- python tools/pmc_step.py` pass (development aid).
python tools/agg_pmc_step.py <out_dir> [iterations]"""
import collections
import csv
import glob
import sys

root = sys.argv[1]
iters = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for path in glob.glob(root + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(path)):
        n = r['Kernel_Name'].replace('void ', '').replace('ldetr::', '')
        n = n.split('(')[0][:70]
        acc[n][r['Counter_Name']] += float(r['Counter_Value']) / iters
        cnt[(n, r['Counter_Name'])] += 1
names = sorted({c for v in acc.values() for c in v})
key = 'SQ_BUSY_CU_CYCLES' if 'SQ_BUSY_CU_CYCLES' in names else names[0]
print(f'{"kernel":72s} launches ' + ' '.join(f'{c[:22]:>22s}' for c in names))
for n, v in sorted(acc.items(), key=lambda kv: -kv[1].get(key, 0))[:40]:
    print(f'{n:72s} {cnt[(n, key)] / iters:8.0f} ' + ' '.join(f'{v.get(c, 0) / 1e6:21.2f}M' for c in names))
