"""Sweep of the tiled kernel's (tile, split-K) choices over the transformer's dense GEMM shapes (development aid; the conv shapes are
tools/sweep_policy.py): runs tools/bench_engine.py under LDETR_DEBUG="FORCE_TILE=..,FORCE_SK=..".  usage: python tools/sweep_policy_gemm.py [batch]"""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B = sys.argv[1] if len(sys.argv) > 1 else '16'
pat = re.compile(r'^(.{28}) M=\s*(\d+) N=\s*(\d+) K=\s*(\d+)\s+fwd\+bias\+relu\s+([\d.]+)us.*?dX\s+([\d.]+)us.*?dW\s+([\d.]+)us')
def run(env):
    e = dict(os.environ); e.update(env); e['LDETR_BENCH_GEMM_ONLY'] = '1'
    out = subprocess.run([sys.executable, os.path.join(root, 'tools', 'bench_engine.py'), B], env=e, capture_output=True, text=True, stdin=subprocess.DEVNULL).stdout
    res = {}
    for line in out.splitlines():
        m = pat.match(line)
        if m:
            res[m.group(1).strip()] = (float(m.group(5)), float(m.group(6)), float(m.group(7)))
    return res
base = run({})
cfgs = [(t, s) for t in (1, 2, 3) for s in (1, 2, 4, 8)]
allr = {c: run({'LDETR_DEBUG': 'FORCE_TILE=%d,FORCE_SK=%d' % c}) for c in cfgs}
tn = {1: '64x64', 2: '128x64', 3: '128x128'}
for name in base:
    line = f'{name:22s}'
    for pi, pn in ((0, 'fwd'), (1, 'dX'), (2, 'dW')):
        best = min(((allr[c][name][pi], c) for c in cfgs if name in allr[c]), default=(0, None))
        line += f' | {pn} policy {base[name][pi]:6.1f}us best {best[0]:6.1f}us ({tn[best[1][0]]} sk{best[1][1]})' if best[1] else ''
    print(line, flush=True)
