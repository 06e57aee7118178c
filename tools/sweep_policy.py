"""Sweep of the tiled kernel's (tile, split-K) choices over the ResNet / StyleGAN conv shapes (development aid): runs
tools/bench_engine.py under LDETR_DEBUG="FORCE_TILE=..,FORCE_SK=.." and prints, per shape and pass, the policy's time next to the best
forced configuration.  usage: python tools/sweep_policy.py [batch]"""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B = sys.argv[1] if len(sys.argv) > 1 else '16'
pat = re.compile(r'^(.{28}) M=\s*(\d+) N=\s*(\d+) K=\s*(\d+)\s+fwd\s+([\d.]+)us.*?bwdD\s+([\d.]+)us.*?bwdW\(sk=\s*\d+\)\s+([\d.]+)us')
def run(env):
    e = dict(os.environ); e.update(env); e['LDETR_BENCH_CONV_ONLY'] = '1'
    out = subprocess.run([sys.executable, os.path.join(root, 'tools', 'bench_engine.py'), B], env=e, capture_output=True, text=True, stdin=subprocess.DEVNULL).stdout
    res = {}
    for line in out.splitlines():
        m = pat.match(line)
        if m:
            res[m.group(1).strip().split(' [')[0]] = (float(m.group(5)), float(m.group(6)), float(m.group(7)))
    return res
base = run({})
cfgs = [(t, s) for t in (1, 2, 3) for s in (1, 2, 4, 8, 16)]
allr = {c: run({'LDETR_DEBUG': 'FORCE_TILE=%d,FORCE_SK=%d' % c}) for c in cfgs}
tn = {1: '64x64', 2: '128x64', 3: '128x128'}
tot_base = [0, 0, 0]; tot_best = [0, 0, 0]
for name in base:
    line = f'{name:26s}'
    for pi, pn in ((0, 'fwd'), (1, 'bwdD'), (2, 'bwdW')):
        best = min(((allr[c][name][pi], c) for c in cfgs if name in allr[c]), default=(0, None))
        tot_base[pi] += base[name][pi]; tot_best[pi] += min(best[0], base[name][pi])
        line += f' | {pn} policy {base[name][pi]:7.1f}us best {best[0]:7.1f}us ({tn[best[1][0]]} sk{best[1][1]})' if best[1] else ''
    print(line, flush=True)
print('sum fwd  policy %.1f us, best %.1f us' % (tot_base[0], tot_best[0]))
print('sum bwdD policy %.1f us, best %.1f us' % (tot_base[1], tot_best[1]))
print('sum bwdW policy %.1f us, best %.1f us' % (tot_base[2], tot_best[2]))
