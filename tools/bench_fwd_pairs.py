"""What would riding an inference-only encoder in the trained encoder's launches buy?  Two INDEPENDENT forward GEMMs of the 64-token encoders
(different operands, same shapes) as one paired small-tile launch vs two launches, hipGraph-replayed (development aid).
Usage: python tools/bench_fwd_pairs.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from layoutdetr_amd.hip import core
from tools.bench_engine import timeit
dev = torch.device('cuda:0')


def case(name, M0, M1, N, K):
    xs = [torch.randn(M, K, device=dev) for M in (M0, M1)]; ws = [torch.randn(N, K, device=dev) for _ in range(2)]; bs = [torch.randn(N, device=dev) for _ in range(2)]
    ys = [torch.empty(M, N, device=dev) for M in (M0, M1)]
    g = [dict(A=xs[i], B=ws[i], ta=0, tb=0, M=(M0, M1)[i], N=N, K=K, out=ys[i], ep=core.epilogue(col_bias=bs[i])) for i in range(2)]
    single = core.gemm_pair_is_single_launch(g[0], g[1])
    tp = timeit(lambda: core.gemm_pair(g[0], g[1]), n=50)
    t0 = timeit(lambda: core.gemm(xs[0], ws[0], 0, 0, M0, N, K, out=ys[0], ep=g[0]['ep']), n=50)
    t1 = timeit(lambda: core.gemm(xs[1], ws[1], 0, 0, M1, N, K, out=ys[1], ep=g[1]['ep']), n=50)
    print(f'{name:28s} M={M0:5d}+{M1:5d} N={N:5d} K={K:5d}  pair {tp*1e6:6.1f}us (one launch: {single})  alone {t0*1e6:6.1f} + {t1*1e6:6.1f} = {(t0+t1)*1e6:6.1f}us', flush=True)


for rows in ((1024, 1024), (2048, 1024), (128, 128), (256, 128)):
    case('qk projection', *rows, 512, 256)
    case('v / out projection', *rows, 256, 256)
    case('linear1', *rows, 2048, 256)
    case('linear2', *rows, 256, 2048)
