"""Reference streaming rates on this GPU (development aid): torch fill / copy / add at the 1x1-conv tensor sizes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = torch.device('cuda:0')
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); g.replay(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / (2 * n) * 1e-3
for mb in (16, 67, 268, 1024):
    n = mb * 1000 * 1000 // 4
    x = torch.randn(n, device=dev); y = torch.empty_like(x); z = torch.randn(n, device=dev)
    t = timeit(lambda: y.fill_(1.0)); print(f'{mb:5d} MB fill  {t*1e6:8.1f} us  write {n*4/t/1e12:5.2f} TB/s')
    t = timeit(lambda: y.copy_(x)); print(f'{mb:5d} MB copy  {t*1e6:8.1f} us  r+w   {2*n*4/t/1e12:5.2f} TB/s')
    t = timeit(lambda: torch.add(x, z, out=y)); print(f'{mb:5d} MB add   {t*1e6:8.1f} us  2r+w  {3*n*4/t/1e12:5.2f} TB/s')
    t = timeit(lambda: x.sum()); print(f'{mb:5d} MB sum   {t*1e6:8.1f} us  read  {n*4/t/1e12:5.2f} TB/s')
