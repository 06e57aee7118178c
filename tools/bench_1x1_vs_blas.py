"""The trunk's 1x1 convolutions as plain GEMMs: engine (no epilogue) vs torch.mm (hipBLASLt / Tensile assembly) on the same shapes and
the same cold/warm regime -- the yardstick for what a 1x1 launch of 2.1 GFLOP can reach at all on this part (development aid).
Usage: python tools/bench_1x1_vs_blas.py [batch]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from layoutdetr_amd.hip import core
from tools.bench_engine import timeit
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
torch.backends.cuda.matmul.allow_tf32 = False
shapes = [('l1 64->64', B * 4096, 64, 64), ('l1 64->256', B * 4096, 256, 64), ('l1 256->64', B * 4096, 64, 256), ('l2 256->128', B * 4096, 128, 256),
          ('l2 128->512', B * 1024, 512, 128), ('l2 512->128', B * 1024, 128, 512), ('l3 256->1024', B * 256, 1024, 256), ('l3 1024->256', B * 256, 256, 1024),
          ('l4 512->2048', B * 64, 2048, 512), ('l4 2048->512', B * 64, 512, 2048)]
for name, M, N, K in shapes:
    # 12 operand sets rotated so that a replayed launch finds its operands in HBM, not in the 256 MB Infinity Cache (the in-step regime)
    nset = max(2, min(12, int(600e6 // ((M * K + M * N) * 4)) + 1))
    As = [torch.randn(M, K, device=dev) for _ in range(nset)]; W = torch.randn(N, K, device=dev); Wt = W.t().contiguous()
    Ys = [torch.empty(M, N, device=dev) for _ in range(nset)]
    fl = 2.0 * M * N * K
    it = [0]
    def eng():
        i = it[0] % nset; it[0] += 1
        core.gemm(As[i], W, 0, 0, M, N, K, out=Ys[i])
    def blas():
        i = it[0] % nset; it[0] += 1
        torch.mm(As[i], Wt, out=Ys[i])
    def dx_eng():
        i = it[0] % nset; it[0] += 1
        core.gemm(Ys[i], W, 0, 1, M, K, N, out=As[i])
    def dx_blas():
        i = it[0] % nset; it[0] += 1
        torch.mm(Ys[i], W, out=As[i])
    dW = torch.empty(N, K, device=dev)
    def dw_eng():
        i = it[0] % nset; it[0] += 1
        core.gemm(Ys[i], As[i], 1, 1, N, K, M, out=dW)
    def dw_blas():
        i = it[0] % nset; it[0] += 1
        torch.mm(Ys[i].t(), As[i], out=dW)
    r = [timeit(f, n=24) for f in (eng, blas, dx_eng, dx_blas, dw_eng, dw_blas)]
    mb = (M * K + M * N + N * K) * 4 / 1e6
    print(f'{name:14s} M={M:6d} N={N:4d} K={K:4d} {mb:6.1f} MB | fwd engine {r[0]*1e6:6.1f}us {fl/r[0]/1e12:6.1f}TF  blas {r[1]*1e6:6.1f}us {fl/r[1]/1e12:6.1f}TF | dX engine {r[2]*1e6:6.1f}us blas {r[3]*1e6:6.1f}us | dW engine {r[4]*1e6:6.1f}us blas {r[5]*1e6:6.1f}us', flush=True)
