"""Development aid: idle time between kernels of the replayed step.
python tools/trace_gaps.py <kernel_trace.csv> [n_last_steps]  (csv from `rocprofv3 --kernel-trace --output-format csv`)
Prints, for the last replayed iterations (delimited by ldetr::ema_kernel launches): wall time, union of kernel intervals (busy),
the idle remainder, the gap histogram, and which kernels are followed by the largest total idle time."""
import collections
import csv
import sys


def main():
    rows = []
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
    rows.sort()
    ema = [i for i, r in enumerate(rows) if 'ema_kernel' in r[2]]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    lo, hi = ema[-n - 1] + 1, ema[-1] + 1
    seg = rows[lo:hi]
    wall = seg[-1][1] - seg[0][0]
    busy = 0; cur_end = seg[0][0]; gaps = []; after = collections.Counter(); overl = 0
    for s, e, name in seg:
        if s > cur_end:
            gaps.append((s - cur_end, prev))
            after[prev[:60]] += s - cur_end
        else:
            overl += min(e, cur_end) - s
        busy += max(0, e - max(s, cur_end))
        if e > cur_end:
            cur_end = e; prev = name
    print(f'{n} iterations: {len(seg)} kernels, wall {wall / n / 1e6:.2f} ms/iter, busy {busy / n / 1e6:.2f}, idle {(wall - busy) / n / 1e6:.2f}, '
          f'sum of durations {sum(e - s for s, e, _ in seg) / n / 1e6:.2f}, overlapped {overl / n / 1e6:.2f}')
    g = sorted(x for x, _ in gaps)
    if g:
        print(f'gaps: {len(g) / n:.0f}/iter  median {g[len(g) // 2] / 1e3:.2f} us  p90 {g[int(len(g) * .9)] / 1e3:.2f}  max {g[-1] / 1e3:.1f}')
        for lo_, hi_ in [(0, 1000), (1000, 2000), (2000, 4000), (4000, 8000), (8000, 1 << 60)]:
            sel = [x for x in g if lo_ <= x < hi_]
            print(f'  {lo_ / 1e3:4.0f}-{hi_ / 1e3 if hi_ < 1 << 59 else float("inf"):4.0f} us: {len(sel) / n:7.1f}/iter  {sum(sel) / n / 1e6:6.2f} ms/iter')
    for name, t in after.most_common(15):
        print(f'  idle after {name:60s} {t / n / 1e6:6.2f} ms/iter')


if __name__ == '__main__':
    main()
