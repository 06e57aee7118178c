"""Micro-benchmark of the f32 MFMA contraction engine on ResNet-50 / DETR / StyleGAN2 shapes (development aid).
Usage: python tools/bench_engine.py [batch]"""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from layoutdetr_amd.hip import core
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16

def timeit(fn, n=20):
    """GPU time per call from a hipGraph replay of n back-to-back launches (no CPU launch gaps in the measurement)."""
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); g.replay(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / (2 * n) * 1e-3

def conv_case(name, N, H, Ci, Co, k, s, p):
    OH = (H + 2 * p - k) // s + 1
    x = torch.randn(N, H, H, Ci, device=dev); w = torch.randn(Co, k, k, Ci, device=dev); dy = torch.randn(N, OH, OH, Co, device=dev)
    y = torch.empty(N, OH, OH, Co, device=dev); dx = torch.empty_like(x); dw = torch.empty_like(w)
    xt = core.tensor4_nhwc(x); dyt = core.tensor4_nhwc(dy)
    fl = 2.0 * N * OH * OH * Co * k * k * Ci
    L = core.lib()
    sc = torch.rand(Co, device=dev) + 0.5; sh = torch.randn(Co, device=dev); res = torch.randn_like(y)
    ep = core.epilogue(col_scale=sc, col_bias=sh, residual=res.reshape(-1, Co), act=core.ACT_RELU)   # FrozenBN + residual + ReLU, as the trunk runs it
    tf = timeit(lambda: L.ldetr_conv2d_fwd_f32(core.ptr(x), ctypes.byref(xt), core.ptr(w), Co, k, k, s, p, core.ptr(y), Co, OH, OH, None, 0, ctypes.byref(ep), core.stream()))
    tb = timeit(lambda: L.ldetr_conv2d_bwd_data_f32(core.ptr(dy), ctypes.byref(dyt), core.ptr(w), Ci, k, k, s, p, core.ptr(dx), Ci, H, H, None, 0, None, core.stream()))
    sk = 0
    tw = timeit(lambda: L.ldetr_conv2d_bwd_weight_f32(core.ptr(x), ctypes.byref(xt), core.ptr(dy), ctypes.byref(dyt), core.ptr(dw), k, k, s, p, sk, None, 0, None, 0, 0, core.stream()))
    xs = torch.rand(N, Ci, device=dev) + 0.5; ds = torch.rand(N, Co, device=dev) + 0.5
    tws = timeit(lambda: L.ldetr_conv2d_bwd_weight_f32(core.ptr(x), ctypes.byref(xt), core.ptr(dy), ctypes.byref(dyt), core.ptr(dw), k, k, s, p, sk, core.ptr(xs), Ci, core.ptr(ds), Co, 0, core.stream())) if name.startswith('sg') else 0
    if tws: name = name + f' [wgrad+scales {tws*1e6:.0f}us {fl/tws/1e12:.1f}TF]'
    print(f'{name:28s} M={N*OH*OH:6d} N={Co:4d} K={k*k*Ci:5d}  fwd {tf*1e6:8.1f}us {fl/tf/1e12:6.1f}TF | bwdD {tb*1e6:8.1f}us {fl/tb/1e12:6.1f}TF | bwdW(sk={sk:3d}) {tw*1e6:8.1f}us {fl/tw/1e12:6.1f}TF', flush=True)
    return fl, tf, tb, tw

def gemm_case(name, M, N, K):
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev); dY = torch.randn(M, N, device=dev); b = torch.randn(N, device=dev)
    fl = 2.0 * M * N * K
    ep = core.epilogue(col_bias=b, act=core.ACT_RELU)
    y = torch.empty(M, N, device=dev); dx = torch.empty(M, K, device=dev); dw = torch.empty(N, K, device=dev)
    t1 = timeit(lambda: core.gemm(A, W, 0, 0, M, N, K, out=y, ep=ep))
    t2 = timeit(lambda: core.gemm(dY, W, 0, 1, M, K, N, out=dx))
    t3 = timeit(lambda: core.gemm(dY, A, 1, 1, N, K, M, out=dw))
    print(f'{name:28s} M={M:6d} N={N:4d} K={K:5d}  fwd+bias+relu {t1*1e6:8.1f}us {fl/t1/1e12:6.1f}TF | dX   {t2*1e6:8.1f}us {fl/t2/1e12:6.1f}TF | dW   {t3*1e6:8.1f}us {fl/t3/1e12:6.1f}TF', flush=True)

def main():
    print("batch", B)
    tot = [0, 0, 0, 0]
    R = 64  # spatial after stem+pool at 256x256
    cases = [('l1 1x1 64->64', B, R, 64, 64, 1, 1, 0), ('l1 3x3 64->64', B, R, 64, 64, 3, 1, 1), ('l1 1x1 64->256', B, R, 64, 256, 1, 1, 0), ('l1 1x1 256->64', B, R, 256, 64, 1, 1, 0),
             ('l2 1x1 256->128', B, R, 256, 128, 1, 1, 0), ('l2 3x3 128->128 s2', B, R, 128, 128, 3, 2, 1), ('l2 1x1 128->512', B, 32, 128, 512, 1, 1, 0), ('l2 1x1 256->512 s2', B, R, 256, 512, 1, 2, 0),
             ('l2 1x1 512->128', B, 32, 512, 128, 1, 1, 0), ('l2 3x3 128->128', B, 32, 128, 128, 3, 1, 1),
             ('l3 3x3 256->256 s2', B, 32, 256, 256, 3, 2, 1), ('l3 1x1 256->1024', B, 16, 256, 1024, 1, 1, 0), ('l3 1x1 1024->256', B, 16, 1024, 256, 1, 1, 0), ('l3 3x3 256->256', B, 16, 256, 256, 3, 1, 1),
             ('l4 3x3 512->512 s2', B, 16, 512, 512, 3, 2, 1), ('l4 1x1 512->2048', B, 8, 512, 2048, 1, 1, 0), ('l4 1x1 2048->512', B, 8, 2048, 512, 1, 1, 0), ('l4 3x3 512->512', B, 8, 512, 512, 3, 1, 1),
             ('sg 3x3 512->512 @16', B, 16, 512, 512, 3, 1, 1), ('sg 3x3 256->256 @32', B, 32, 256, 256, 3, 1, 1), ('sg 3x3 128->128 @64', B, 64, 128, 128, 3, 1, 1),
             ('sg 3x3 64->64 @128', B, 128, 64, 64, 3, 1, 1), ('sg 3x3 32->32 @256', B, 256, 32, 32, 3, 1, 1)]
    if os.environ.get('LDETR_BENCH_GEMM_ONLY'):
        cases = []
    if os.environ.get('LDETR_BENCH_ONLY'):
        cases = [c for c in cases if c[0].startswith(os.environ['LDETR_BENCH_ONLY'])]
    for c in cases:
        conv_case(*c)
    if os.environ.get('LDETR_BENCH_CONV_ONLY'):
        return
    for g in [('enc proj 256', B * 64, 256, 256), ('encdec proj 256', B * 80, 256, 256), ('dec proj 256', B * 9, 256, 256), ('dec proj 256 (10)', B * 10, 256, 256), ('enc qk proj', B * 64, 512, 256), ('enc ffn1', B * 64, 2048, 256), ('enc ffn2', B * 64, 256, 2048), ('dec ffn1', B * 9, 2048, 256), ('dec ffn2', B * 9, 256, 2048), ('dec ffn2 (10)', B * 10, 256, 2048), ('fc_in 3072->768', B * 9, 768, 3072), ('mapping 512', B, 512, 512)]:
        gemm_case(*g)


if __name__ == '__main__':
    main()
