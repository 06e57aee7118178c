"""The paired small-tile launches of the transformer backward (data + weight gradient of one projection; the feed-forward block's two
weight gradients) at the step's shapes, hipGraph-replayed (development aid).  Usage: python tools/bench_small_pairs.py [batch]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from layoutdetr_amd.hip import core
from tools.bench_engine import timeit
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16


def lin_pair(name, M, N, K):
    """backward of y[M,N] = x[M,K] W[N,K]^T: dX = dY W (NN) + dW += dY^T X (TN, bias row sums)"""
    dY = torch.randn(M, N, device=dev); W = torch.randn(N, K, device=dev); X = torch.randn(M, K, device=dev)
    dX = torch.empty(M, K, device=dev); dW = torch.zeros(N, K, device=dev); db = torch.zeros(N, device=dev)
    def f():
        core.gemm_pair(dict(A=dY, B=W, ta=0, tb=1, M=M, N=K, K=N, out=dX, ep=core.epilogue()),
                       dict(A=dY, B=X, ta=1, tb=1, M=N, N=K, K=M, out=dW, ep=core.epilogue(accumulate=True, a_rowsum=db)))
    def f0():
        core.gemm(dY, W, 0, 1, M, K, N, out=dX)
    def f1():
        core.gemm(dY, X, 1, 1, N, K, M, out=dW, ep=core.epilogue(accumulate=True, a_rowsum=db))
    t, t0, t1 = timeit(f, n=50), timeit(f0, n=50), timeit(f1, n=50)
    fl = 4.0 * M * N * K
    print(f'{name:34s} M={M:5d} N={N:5d} K={K:5d}  pair {t*1e6:6.1f}us {fl/t/1e12:5.1f}TF | dX alone {t0*1e6:6.1f}us | dW alone {t1*1e6:6.1f}us', flush=True)


def ffn_pair(name, M):
    dr = torch.randn(M, 256, device=dev); h = torch.randn(M, 2048, device=dev); dh = torch.randn(M, 2048, device=dev); x1 = torch.randn(M, 256, device=dev)
    dW2 = torch.zeros(256, 2048, device=dev); db2 = torch.zeros(256, device=dev); dW1 = torch.zeros(2048, 256, device=dev); db1 = torch.zeros(2048, device=dev)
    def f():
        core.gemm_pair(dict(A=dr, B=h, ta=1, tb=1, M=256, N=2048, K=M, out=dW2, ep=core.epilogue(accumulate=True, a_rowsum=db2)),
                       dict(A=dh, B=x1, ta=1, tb=1, M=2048, N=256, K=M, out=dW1, ep=core.epilogue(accumulate=True, a_rowsum=db1)))
    t = timeit(f, n=50)
    fl = 8.0 * M * 256 * 2048
    print(f'{name:34s} M={M:5d}                    pair {t*1e6:6.1f}us {fl/t/1e12:5.1f}TF', flush=True)


def fwd(name, M, N, K):
    X = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev); y = torch.empty(M, N, device=dev)
    t = timeit(lambda: core.gemm(X, W, 0, 0, M, N, K, out=y, ep=core.epilogue(col_bias=b)), n=50)
    print(f'{name:34s} M={M:5d} N={N:5d} K={K:5d}  fwd  {t*1e6:6.1f}us {2.0*M*N*K/t/1e12:5.1f}TF', flush=True)


print('batch', B)
for L in (9, 10):
    for mult in (1, 2):
        M = B * L * mult
        lin_pair(f'in_proj 256->768 ({L} tok x {B*mult})', M, 768, 256)
        lin_pair(f'out_proj 256->256 ({L} tok x {B*mult})', M, 256, 256)
        ffn_pair(f'ffn weight gradients ({L} tok x {B*mult})', M)
        fwd(f'q projection ({L} tok x {B*mult})', M, 256, 256)
for mult in (1, 2):
    M = B * 64 * mult
    lin_pair(f'enc proj 256->256 (64 tok x {B*mult})', M, 256, 256)
    lin_pair(f'enc qk proj 256->512 (64 tok x {B*mult})', M, 512, 256)
    fwd(f'enc qk proj fwd', M, 512, 256)
    fwd(f'kv grouped fwd 256->1536', M, 1536, 256)
lin_pair('fc_in 3072->768', B * 9, 768, 3072)
lin_pair('mlp 768->768', B * 9, 768, 768)
lin_pair('mlp 768->256', B * 9, 256, 768)
