"""Timing of the device background preprocessing against its HBM roofline, with the oracle (numpy restatement of
Pillow's resize) and Pillow itself timed beside it on the host.  usage: python tools/bench_resample.py [n] [H] [S]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from layoutdetr_amd.training.dataset_layoutganpp import background_to_tensor
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
H = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
S = int(sys.argv[3]) if len(sys.argv) > 3 else 256
dev = torch.device('cuda:0')
imgs = torch.randint(0, 256, (n, H, H, 3), dtype=torch.uint8, device=dev)
for _ in range(3):
    out = background_to_tensor(imgs, S)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(10):
        out = background_to_tensor(imgs, S)
g.replay(); torch.cuda.synchronize()
s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
s.record(); g.replay(); e.record(); torch.cuda.synchronize()
us = s.elapsed_time(e) / 10 * 1e3
alg = n * (H * H * 3 + 2 * H * S * 3 + S * S * 3 * 4)   # read pages, write + read the uint8 intermediate, write floats
print(f'{n} pages {H}x{H} -> {S}: {us:.1f} us per batch, {n / us * 1e6:.0f} pages/s, algorithmic {alg / 1e6:.1f} MB -> {alg / us / 1e6:.2f} TB/s of ~8 TB/s HBM')
try:
    from PIL import Image
    a = imgs[0].cpu().numpy(); t = time.time()
    for _ in range(5): np.array(Image.fromarray(a).resize((S, S), Image.LANCZOS))
    print(f'Pillow {Image.__version__ if hasattr(Image, "__version__") else ""} on one host core: {(time.time() - t) / 5 * 1e3:.2f} ms per page')
except ImportError:
    pass
