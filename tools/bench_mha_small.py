"""The one-launch self-attention forward stand-alone, hipGraph-replayed (development aid).  Usage: python tools/bench_mha_small.py [batch] [L]"""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from layoutdetr_amd.hip import core
from tools.bench_engine import timeit
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
L = int(sys.argv[2]) if len(sys.argv) > 2 else 10
d, H = 256, 8
M = B * L
x = torch.randn(M, d, device=dev); Wi = torch.randn(3 * d, d, device=dev) * 0.05; bi = torch.randn(3 * d, device=dev); Wo = torch.randn(d, d, device=dev) * 0.05
qkv = torch.empty(M, 3 * d, device=dev); o = torch.empty(M, d, device=dev); lse = torch.empty(B * H * L, device=dev); yp = torch.empty(H, M, d, device=dev)
kpm = torch.zeros(B, L, dtype=torch.uint8, device=dev)
for p_drop in (0.0, 0.1):
    f = lambda: core.check(core.lib().ldetr_mha_small_fwd_f32(core.ptr(x), d, core.ptr(Wi), core.ptr(bi), core.ptr(Wo), core.ptr(kpm), core.ptr(qkv), core.ptr(o), core.ptr(lse),
                                                              core.ptr(yp), B, L, d, H, 1.0 / math.sqrt(32), p_drop, 1234, core.seed_ptr() if p_drop > 0 else None, core.stream()), 'mha')
    t = timeit(f, n=50)
    print(f'mha_small fwd B={B} L={L} p_drop={p_drop}: {t*1e6:.2f} us', flush=True)
