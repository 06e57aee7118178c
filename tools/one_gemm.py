"""Runs ONE dense GEMM shape repeatedly (for rocprofv3 --pmc).  Usage: python tools/one_gemm.py M N K [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from layoutdetr_amd.hip import core
dev = torch.device('cuda:0')
M, N, K = [int(v) for v in sys.argv[1:4]]; reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev); y = torch.empty(M, N, device=dev)
for _ in range(reps):
    core.gemm(A, W, 0, 0, M, N, K, out=y)
torch.cuda.synchronize()
