"""One plane-format conv launch x 10 for `rocprofv3 --pmc` (development): python tools/p3_pmc.py fwd|bwdd|bwdw [N H Ci Co k s pad]"""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from layoutdetr_amd.hip import core
from layoutdetr_amd import _lib
sys.argv = [sys.argv[0]] + sys.argv[1:]
import importlib.util
spec = importlib.util.spec_from_file_location('p3_dev', os.path.join(os.path.dirname(os.path.abspath(__file__)), 'p3_dev.py'))
d = importlib.util.module_from_spec(spec); spec.loader.exec_module(d)
what = sys.argv[1] if len(sys.argv) > 1 else 'fwd'
N, H, Ci, Co, k, s, pad = [int(v) for v in sys.argv[2:9]] if len(sys.argv) >= 9 else (16, 32, 128, 128, 3, 1, 1)
dev = d.dev; L = d.L
OH = (H + 2 * pad - k) // s + 1
x = torch.randn(N, H, H, Ci, device=dev); w = torch.randn(Co, k, k, Ci, device=dev) / (k * Ci ** 0.5); dy = torch.randn(N, OH, OH, Co, device=dev)
sc = torch.rand(Co, device=dev) + 0.5
xp = d.p3_split(x.reshape(-1, Ci)); wp = d.p3_split(w.reshape(Co, -1)); dyp = d.p3_split(dy.reshape(-1, Co)); wb = d.p3_weight_bwd(w, sc)
yp = torch.empty(N * OH * OH * Co * 6, dtype=torch.uint8, device=dev); dxp = torch.empty(N * H * H * Ci * 6, dtype=torch.uint8, device=dev)
dw = torch.zeros(Co, k, k, Ci, device=dev)
ep = _lib.P3Epilogue(); ep.alpha = 1.0; ep.col_scale = sc.data_ptr(); ep.relu = 1
for _ in range(10):
    if what == 'fwd':
        d.run_fwd(xp, N, H, Ci, wp, Co, k, s, pad, ep, yp, None)
    elif what == 'bwdd':
        L.ldetr_p3_conv2d_bwd_data(core.ptr(dyp), N, OH, OH, Co, core.ptr(wb), Ci, k, k, s, pad, H, H, None, core.ptr(dxp), None, core.stream())
    else:
        L.ldetr_p3_conv2d_bwd_weight(core.ptr(xp), N, H, H, Ci, core.ptr(dyp), Co, k, k, s, pad, core.ptr(sc), core.ptr(dw), core.stream())
torch.cuda.synchronize()
