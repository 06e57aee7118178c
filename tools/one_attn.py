"""Runs the DETR encoder self-attention shape fwd+bwd repeatedly (for rocprofv3 --pmc / --stats)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from layoutdetr_amd.hip.attention import _AttnPackedFn
dev = torch.device('cuda:0')
B, H, L, dh = 16, 8, int(sys.argv[1]) if len(sys.argv) > 1 else 64, 32
qkv = torch.randn(B * L, 3 * H * dh, device=dev, requires_grad=True)
g = torch.randn(B * L, H * dh, device=dev)
for _ in range(5):
    out = _AttnPackedFn.apply(qkv, None, None, B, H, L, 0.1, False)
    out.backward(g)
torch.cuda.synchronize()
