"""One ResNet 3x3 convolution (layer2 shape at 16 samples: 16 x 32 x 32 x 128 -> 128, forward) launched 20 times on the f32 MFMA pipe or on
the bf16 split pipe: the process `rocprofv3 --pmc ...` wraps to read the issue / busy counters of the engine's k-loop.
  LDETR_DEBUG=SPLIT_BF16=0|7 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d out -- python tools/pmc_conv.py [bwd_data|bwd_weight]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from layoutdetr_amd.hip import core

dev = torch.device('cuda:0')
what = sys.argv[1] if len(sys.argv) > 1 else 'fwd'
N, H, C, O = 16, 32, 128, 128
x = torch.randn(N, H, H, C, device=dev); w = torch.randn(O, 3, 3, C, device=dev); dy = torch.randn(N, H, H, O, device=dev)
y = torch.empty(N, H, H, O, device=dev); dx = torch.empty_like(x); dw = torch.empty_like(w)
xt = core.tensor4_nhwc(x); dyt = core.tensor4_nhwc(dy)
L = core.lib()
for _ in range(20):
    if what == 'fwd':
        L.ldetr_conv2d_fwd_f32(core.ptr(x), ctypes.byref(xt), core.ptr(w), O, 3, 3, 1, 1, core.ptr(y), O, H, H, None, 0, None, core.stream())
    elif what == 'bwd_data':
        L.ldetr_conv2d_bwd_data_f32(core.ptr(dy), ctypes.byref(dyt), core.ptr(w), C, 3, 3, 1, 1, core.ptr(dx), C, H, H, None, 0, None, core.stream())
    else:
        L.ldetr_conv2d_bwd_weight_f32(core.ptr(x), ctypes.byref(xt), core.ptr(dy), ctypes.byref(dyt), core.ptr(dw), 3, 3, 1, 1, 0, None, 0, None, 0, 0, core.stream())
torch.cuda.synchronize()
