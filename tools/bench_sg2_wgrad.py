"""Weight gradient of the StyleGAN2 synthesis layers: per-sample operand scales inside the launch (sample-aligned K slices + atomics) vs operands scaled
beforehand (two element-wise launches) and a plain weight-gradient launch.  python tools/bench_sg2_wgrad.py [B]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from layoutdetr_amd.hip import core  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device('cuda:0')
LAYERS = [(4, 512, 512, 1), (4, 512, 512, 2), (8, 512, 512, 1), (8, 512, 512, 2), (16, 512, 512, 1), (16, 512, 256, 2), (32, 256, 256, 1), (32, 256, 128, 2),
          (64, 128, 128, 1), (64, 128, 64, 2), (128, 64, 64, 1), (128, 64, 32, 2)]


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for R, I, O, up in LAYERS:
    x = torch.randn(B, R, R, I, device=dev)
    OH = R if up == 1 else 2 * R + 1
    dv = torch.randn(B, OH, OH, O, device=dev)
    s = torch.randn(B, I, device=dev) + 1; d = torch.rand(B, O, device=dev) + 0.5
    dw = torch.zeros(O, 3, 3, I, device=dev)
    xt, dvt = core.tensor4_nhwc(x), core.tensor4_nhwc(dv)
    fn = core.lib().ldetr_conv2d_bwd_weight_f32 if up == 1 else core.lib().ldetr_conv_transpose2d_bwd_weight_f32
    st, pad = (1, 1) if up == 1 else (2, 0)

    def scaled():
        core.check(fn(core.ptr(x), ctypes.byref(xt), core.ptr(dv), ctypes.byref(dvt), core.ptr(dw), 3, 3, st, pad, 0, core.ptr(s), s.stride(0), core.ptr(d), d.stride(0), 1, core.stream()), 'a')

    def pre():
        xs = x * s[:, None, None, :]; dvd = dv * d[:, None, None, :]
        core.check(fn(core.ptr(xs), ctypes.byref(xt), core.ptr(dvd), ctypes.byref(dvt), core.ptr(dw), 3, 3, st, pad, 0, None, 0, None, 0, 1, core.stream()), 'b')
    dw.zero_(); scaled(); r0 = dw.clone(); k0 = core.launched_kernel('gemm')
    dw.zero_(); pre(); r1 = dw.clone(); k1 = core.launched_kernel('gemm')
    err = ((r0 - r1).abs().max() / r0.abs().max()).item()
    t0, t1 = timeit(scaled), timeit(pre)
    gf = 2.0 * B * R * R * O * 9 * I / 1e9
    print(f'B={B} R={R:3d} {I:3d}->{O:3d} up={up}: in-launch scales {t0:7.1f} us ({gf / t0 * 1e3:6.1f} TF) [{k0}] | pre-scaled {t1:7.1f} us ({gf / t1 * 1e3:6.1f} TF) [{k1}]  diff {err:.1e}')
