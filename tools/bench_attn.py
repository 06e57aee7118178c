"""Stand-alone times of the fused attention kernels at the step's shapes (hipGraph-replayed).  python tools/bench_attn.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from bench_hbm_kernels import timed
from layoutdetr_amd.hip.attention import _AttnFn, _AttnPackedFn
dev = torch.device('cuda')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
H, dh = 8, 32
for name, Lq, Lk, packed in (('encoder self S=64', 64, 64, True), ('decoder self Lq=10', 10, 10, True), ('decoder cross 10x64', 10, 64, False), ('encoder self S=256', 256, 256, True)):
    d = H * dh
    if packed:
        def prep():
            qkv = torch.randn(B * Lq, 3 * d, device=dev, requires_grad=True)
            return qkv
        qkv = torch.randn(B * Lq, 3 * d, device=dev)
        with torch.no_grad():
            tf = timed(lambda: _AttnPackedFn.apply(qkv, None, None, B, H, Lq, 0.0))
        def prepb():
            x = qkv.clone().requires_grad_(True)
            o = _AttnPackedFn.apply(x, None, None, B, H, Lq, 0.0)
            g = torch.randn_like(o)
            return lambda: torch.autograd.grad(o, x, g, retain_graph=True)
        tb = timed(None, prepare=prepb)
    else:
        q = torch.randn(B * Lq, d, device=dev); k = torch.randn(B * Lk, d, device=dev); v = torch.randn(B * Lk, d, device=dev)
        with torch.no_grad():
            tf = timed(lambda: _AttnFn.apply(q, k, v, None, B, H, Lq, Lk, 0.0))
        def prepb():
            qq, kk, vv = (t.clone().requires_grad_(True) for t in (q, k, v))
            o = _AttnFn.apply(qq, kk, vv, None, B, H, Lq, Lk, 0.0)
            g = torch.randn_like(o)
            return lambda: torch.autograd.grad(o, (qq, kk, vv), g, retain_graph=True)
        tb = timed(None, prepare=prepb)
    fl = 4.0 * Lq * Lk * dh * B * H
    print(f'{name:24s} B={B}: fwd {tf * 1e6:7.1f} us ({fl / tf / 1e12:6.2f} TF)  bwd {tb * 1e6:7.1f} us', flush=True)
