"""Runs ONE engine shape repeatedly (for rocprofv3 --pmc).  Usage: python tools/one_kernel.py fwd|bwdd|bwdw N H Ci Co k s p [reps]"""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from layoutdetr_amd.hip import core
dev = torch.device('cuda:0')
mode = sys.argv[1]; N, H, Ci, Co, k, s, p = [int(v) for v in sys.argv[2:9]]; reps = int(sys.argv[9]) if len(sys.argv) > 9 else 10
OH = (H + 2 * p - k) // s + 1
x = torch.randn(N, H, H, Ci, device=dev); w = torch.randn(Co, k, k, Ci, device=dev); dy = torch.randn(N, OH, OH, Co, device=dev)
y = torch.empty(N, OH, OH, Co, device=dev); dx = torch.empty_like(x); dw = torch.empty_like(w)
xt = core.tensor4_nhwc(x); dyt = core.tensor4_nhwc(dy); L = core.lib(); st = core.stream()
for _ in range(reps):
    if mode == 'fwd':
        L.ldetr_conv2d_fwd_f32(core.ptr(x), ctypes.byref(xt), core.ptr(w), Co, k, k, s, p, core.ptr(y), Co, OH, OH, None, 0, None, st)
    elif mode == 'bwdd':
        L.ldetr_conv2d_bwd_data_f32(core.ptr(dy), ctypes.byref(dyt), core.ptr(w), Ci, k, k, s, p, core.ptr(dx), Ci, H, H, None, 0, None, st)
    else:
        sk = 0
        L.ldetr_conv2d_bwd_weight_f32(core.ptr(x), ctypes.byref(xt), core.ptr(dy), ctypes.byref(dyt), core.ptr(dw), k, k, s, p, sk, None, 0, None, 0, 0, st)
torch.cuda.synchronize()
