"""Development aid: the ORDERED kernel sequence of one replayed iteration (what follows what on the phase's in-order queue).
python tools/trace_order.py <kernel_trace.csv> [out.txt]   (csv from `rocprofv3 --kernel-trace --output-format csv`)
One line per launch of the last complete iteration (delimited by adam_kernel<true> = G's optimiser step with the EMA ride-along):
start offset [us], duration [us], gap to the previous kernel's end [us], short kernel name.  A run-length summary follows."""
import collections
import csv
import re
import sys


def short(name):
    name = re.sub(r'^void ', '', name)
    name = name.replace('ldetr::', '').replace('at::native::', 'at::')
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    m = re.match(r'([^(]*?)(<.*>)?\(', name)
    if m:
        base, targs = m.group(1), m.group(2) or ''
        if base.startswith('at::') or 'rocclr' in base:
            f = re.search(r'(\w+Functor\w*|direct_copy|CatArray\w*|reduce_kernel|gather|index\w*|embedding\w*|distribution\w*)', name)
            return 'at::' + (f.group(1) if f else base[4:40])
        return base + (targs if len(targs) < 40 else targs[:40] + '>')
    return name[:70]


def main():
    rows = []
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if 'adam_kernel<true>' in r[2]]
    lo, hi = marks[-2] + 1, marks[-1] + 1
    seg = rows[lo:hi]
    out = open(sys.argv[2], 'w') if len(sys.argv) > 2 else sys.stdout
    t0 = seg[0][0]
    prev_end = t0
    agg = collections.OrderedDict()
    for s, e, name in seg:
        sn = short(name)
        out.write(f'{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} {(s - prev_end) / 1e3:6.1f}  {sn}\n')
        prev_end = max(prev_end, e)
        a = agg.setdefault(sn, [0, 0.0])
        a[0] += 1; a[1] += (e - s) / 1e3
    out.write(f'# {len(seg)} launches, wall {(seg[-1][1] - t0) / 1e6:.3f} ms, sum of durations {sum(e - s for s, e, _ in seg) / 1e6:.3f} ms\n')
    for sn, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        out.write(f'# {c:5d} {t / 1e3:8.3f} ms  {t / c:7.1f} us  {sn}\n')


if __name__ == '__main__':
    main()
