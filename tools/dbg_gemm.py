import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from layoutdetr_amd.hip import core
from tools.bench_engine import timeit
dev = torch.device('cuda:0')
for (M, N, K) in [(4096, 2304, 768), (8192, 2304, 768), (4096, 2048, 768), (4096, 4096, 768), (4096, 4096, 1024), (4096, 2048, 1024), (4096, 1024, 768), (4096, 1536, 768), (4096, 2048, 512), (4096, 2048, 256+128)]:
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev); y = torch.empty(M, N, device=dev)
    fl = 2.0 * M * N * K
    t0 = timeit(lambda: core.gemm(A, W, 0, 0, M, N, K, out=y), n=5)
    t1 = timeit(lambda: core.gemm(A, W, 0, 0, M, N, K, out=y, ep=core.epilogue(col_bias=b)), n=5)
    t2 = timeit(lambda: core.gemm(A, W, 0, 0, M, N, K, out=y, ep=core.epilogue(col_bias=b, act=core.ACT_GELU)), n=5)
    t3 = timeit(lambda: torch.mm(A, W.t(), out=y), n=5)
    print(f'M={M} N={N} K={K}: plain {t0*1e6:7.0f}us {fl/t0/1e12:5.1f}TF | +bias {t1*1e6:7.0f}us | +bias+gelu {t2*1e6:7.0f}us | hipBLASLt {t3*1e6:7.0f}us {fl/t3/1e12:5.1f}TF', flush=True)
