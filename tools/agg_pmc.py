import csv, glob, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(float); n = collections.Counter()
for path in glob.glob(root + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(path)):
        if 'gemm_f32_kernel' not in r['Kernel_Name']: continue
        acc[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
print(root, {k: round(v / n[k]) for k, v in sorted(acc.items())}, 'launches', max(n.values()) if n else 0)
