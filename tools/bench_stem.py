"""Time of the ResNet stem forward (7x7 / 2, NCHW image in, FrozenBN + ReLU fused) at the bench's size (development aid)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from layoutdetr_amd.hip import conv

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
x = torch.randn(B, 3, 256, 256, device='cuda'); w = torch.randn(64, 3, 7, 7, device='cuda').contiguous(memory_format=torch.channels_last)
sc = torch.rand(64, device='cuda') + 0.5; sh = torch.randn(64, device='cuda')


def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); g.replay(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / (2 * n) * 1e-3


t = timeit(lambda: conv.conv2d_nhwc(x, w, sc, sh, None, stride=2, pad=3, relu=True, x_is_nchw=True))
print(f'stem fwd B={B}: {t * 1e6:.1f} us  {2 * B * 128 * 128 * 64 * 147 / t / 1e12:.1f} TFLOP/s')
