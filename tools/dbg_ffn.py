"""Development aid: run-to-run determinism of the fused feed-forward block (forward bitwise, backward up to atomic order)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from layoutdetr_amd.hip import ffn  # noqa: E402

dev = torch.device('cuda:0')
torch.manual_seed(1)
D, F = 256, 2048
for M, pos, wg in [(18, False, True), (8, True, True), (20, False, False), (144, False, True), (320, False, True)]:
    l1 = torch.nn.Linear(D, F).to(dev); l2 = torch.nn.Linear(F, D).to(dev); ln = torch.nn.LayerNorm(D).to(dev)
    if not wg:
        for m in (l1, l2, ln):
            m.requires_grad_(False)
    x = torch.randn(M, D, device=dev); gy = torch.randn(M, D, device=dev); rres = torch.randn(M, D, device=dev); lna = torch.nn.LayerNorm(D).to(dev)
    P = torch.randn(M // 2, D, device=dev) if pos else None
    outs = []
    for rep in range(3):
        for m in (l1, l2, ln):
            for p_ in m.parameters():
                p_.grad = None
        xx = x.clone().requires_grad_(True)
        o = ffn.add_ln_ffn_add_ln(xx, rres, lna, 0.0, l1, l2, ln, 0.0, 0.0, pos=P)
        y = o[0] if pos else o
        loss = (y * gy).sum() + ((o[1] * gy).sum() if pos else 0)
        loss.backward()
        outs.append((y.detach().clone(), xx.grad.clone(), [p_.grad.clone() for m in (l1, l2, ln) for p_ in m.parameters() if p_.grad is not None]))
    for rep in (1, 2):
        dy = (outs[rep][0] - outs[0][0]).abs().max().item()
        dx = ((outs[rep][1] - outs[0][1]).abs().max() / outs[0][1].abs().max()).item()
        dw = max([((a - b).abs().max() / b.abs().max()).item() for a, b in zip(outs[rep][2], outs[0][2])] + [0.0])
        print(f'M={M} pos={pos} wgrad={wg} rep{rep}: y diff {dy:.2e}  dx rel diff {dx:.2e}  worst param-grad rel diff {dw:.2e}')
