"""Prototype harness: fp32 GEMM on the bf16 matrix pipe with a 3-way operand split (see gemm_bf16x6.hip) against the product
library's f32-MFMA engine: accuracy vs an fp64 reference, and time per launch from a hipGraph replay.
Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared tools/proto_bf16x6/gemm_bf16x6.hip -o tools/proto_bf16x6/libproto_bf16x6.so
Run:   python tools/proto_bf16x6/run.py"""
import ctypes
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import torch
from layoutdetr_amd.hip import core

dev = torch.device('cuda:0')
P = ctypes.CDLL(os.path.join(HERE, 'libproto_bf16x6.so'))
P.proto_gemm_bf16_split.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_long,
                                    ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]


def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); g.replay(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / (2 * n) * 1e-3


def split(A, B, C, nprod):
    M, K = A.shape; N = B.shape[0]
    rc = P.proto_gemm_bf16_split(A.data_ptr(), K, B.data_ptr(), K, C.data_ptr(), N, M, N, K, nprod, torch.cuda.current_stream().cuda_stream)
    assert rc == 0


def main():
    torch.manual_seed(0)
    core.lib()
    for (M, N, K) in [(4096, 4096, 4096), (8192, 2048, 1024), (16384, 512, 1152), (65536, 256, 256), (65536, 128, 576), (1024, 2048, 256), (4096, 512, 4608)]:
        A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev)
        # spread of magnitudes: a column scale so that cancellation and mixed exponents are exercised
        A *= torch.exp(torch.randn(1, K, device=dev)); B *= torch.exp(torch.randn(1, K, device=dev) * 0.5)
        ref = (A[:1024].double() @ B.double().t())
        den = ref.abs().max().item()
        out = {}
        for name, nprod in [('bf16x6', 6), ('bf16x3', 3), ('bf16x1', 1)]:
            C = torch.empty(M, N, device=dev)
            split(A, B, C, nprod)
            err = (C[:1024].double() - ref).abs()
            t = timeit(lambda: split(A, B, C, nprod))
            out[name] = (err.max().item() / den, err.pow(2).mean().sqrt().item() / den, t)
        C = torch.empty(M, N, device=dev)
        core.gemm(A, B, 0, 0, M, N, K, out=C)
        err = (C[:1024].double() - ref).abs()
        t = timeit(lambda: core.gemm(A, B, 0, 0, M, N, K, out=C))
        out['f32 engine'] = (err.max().item() / den, err.pow(2).mean().sqrt().item() / den, t)
        Ct = A @ B.t()
        err = (Ct[:1024].double() - ref).abs()
        t = timeit(lambda: torch.matmul(A, B.t(), out=Ct))
        out['hipBLASLt f32'] = (err.max().item() / den, err.pow(2).mean().sqrt().item() / den, t)
        fl = 2.0 * M * N * K
        print(f'M={M} N={N} K={K}')
        for k, (emax, erms, t) in out.items():
            print(f'   {k:14s} max err {emax:.2e}  rms err {erms:.2e}   {t * 1e6:8.1f} us  {fl / t / 1e12:6.1f} TFLOP/s (fp32-equivalent)', flush=True)


if __name__ == '__main__':
    main()
