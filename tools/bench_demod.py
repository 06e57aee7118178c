"""Timing of the fused demodulation kernels (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from layoutdetr_amd.hip import core
dev = torch.device('cuda:0')
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
L = core.lib()
for (B, O, I, K) in [(16, 512, 512, 3), (16, 256, 256, 3), (16, 64, 64, 3), (16, 32, 32, 3)]:
    w = torch.randn(O, I, K, K, device=dev).contiguous(memory_format=torch.channels_last); s = torch.randn(B, I, device=dev)
    d = torch.empty(B, O, device=dev); w2 = torch.empty(O, I, device=dev); g = torch.randn(B, O, device=dev)
    dw = torch.zeros_like(w); ds = torch.empty_like(s); st = w.stride()
    f = timeit(lambda: L.ldetr_demod_fwd_f32(core.ptr(w), st[0], st[1], st[2], st[3], core.ptr(s), core.ptr(d), core.ptr(w2), B, O, I, K, K, 1e-8, core.stream()))
    bw = timeit(lambda: L.ldetr_demod_bwd_f32(core.ptr(w), st[0], st[1], st[2], st[3], core.ptr(s), core.ptr(d), core.ptr(w2), core.ptr(g), core.ptr(dw), 1, None, B, O, I, K, K, core.stream()))
    bs = timeit(lambda: L.ldetr_demod_bwd_f32(core.ptr(w), st[0], st[1], st[2], st[3], core.ptr(s), core.ptr(d), core.ptr(w2), core.ptr(g), None, 0, core.ptr(ds), B, O, I, K, K, core.stream()))
    print(f'B={B} O={O} I={I} K={K}: fwd {f:.1f} us, bwd weight {bw:.1f} us, bwd styles {bs:.1f} us', flush=True)
