"""Bandwidth of the fused bias_act-backward + reductions kernel (rowreduce_kernel<ACTGRAD>) on the StyleGAN2 decoder's shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from layoutdetr_amd.hip import core
dev = torch.device('cuda:0')
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3
for (B, R, C) in [(16, 256, 32), (16, 128, 64), (16, 64, 128), (16, 32, 256), (16, 16, 512), (1, 144, 2048), (1, 1024, 2048)]:
    P = R * R if B > 1 else R
    dy = torch.randn(B * P, C, device=dev); y = torch.randn(B * P, C, device=dev); dv = torch.empty_like(dy)
    bias = torch.randn(C, device=dev); demod = torch.rand(B, C, device=dev) + 0.5
    db = torch.zeros(C, device=dev); dd = torch.zeros(B, C, device=dev)
    L = core.lib()
    t = timeit(lambda: L.ldetr_act_bwd_reduce_f32(core.ptr(dy), core.ptr(y), core.ptr(dv), core.ptr(bias), core.ptr(demod), core.ptr(db), core.ptr(dd), B, P, C, 2, 0.2, 1.414, core.stream()))
    byt = dy.numel() * 12
    print(f'B={B:2d} P={P:6d} C={C:4d}: {t*1e6:8.1f} us  {byt/1e6:7.1f} MB  {byt/t/1e12:5.2f} TB/s', flush=True)
