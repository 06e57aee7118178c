"""Large dense GEMMs through the engine: the core loop's ceiling without gather addressing (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from layoutdetr_amd.hip import core
from tools.bench_engine import timeit
dev = torch.device('cuda:0')
for (M, N, K) in [(8192, 4096, 4096), (4096, 4096, 1024), (16384, 1024, 1024), (65536, 256, 1152)]:
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev); Bt = torch.randn(K, N, device=dev); At = torch.randn(K, M, device=dev)
    y = torch.empty(M, N, device=dev)
    fl = 2.0 * M * N * K
    t1 = timeit(lambda: core.gemm(A, W, 0, 0, M, N, K, out=y), n=5)
    t2 = timeit(lambda: core.gemm(A, Bt, 0, 1, M, N, K, out=y), n=5)
    t3 = timeit(lambda: core.gemm(At, Bt, 1, 1, M, N, K, out=y), n=5)
    t4 = timeit(lambda: torch.mm(A, Bt, out=y), n=5)
    print(f'M={M} N={N} K={K}: NT {fl/t1/1e12:6.1f} TF | NN {fl/t2/1e12:6.1f} TF | TN {fl/t3/1e12:6.1f} TF | torch.mm (hipBLASLt) {fl/t4/1e12:6.1f} TF', flush=True)
