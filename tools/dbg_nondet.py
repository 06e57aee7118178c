"""Development aid: run-to-run differences of one iteration's gradients (B=2, 64x64, eval), per parameter, for two identical runs."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_model_gpu as T  # noqa: E402

dev = torch.device('cuda:0')
res = []
for rep in range(2):
    g, fm = T._iteration_grads(dev, 'plain')
    res.append((g, fm))
for k in ('Gmain', 'Dmain'):
    a, b = res[0][0][k][0], res[1][0][k][0]
    mx = a.abs().max().item()
    print(k, 'flat max', mx, 'max abs diff', (a - b).abs().max().item(), 'rel', ((a - b).abs().max() / mx).item())
    fm = res[0][1] if k == 'Gmain' else None
    if fm is None:
        continue
    rows = []
    for n, off, p in zip(fm.names, fm.offsets, fm.params):
        sl = slice(off, off + p.numel())
        d = (a[sl] - b[sl]).abs().max().item()
        rows.append((d, n, a[sl].abs().max().item()))
    rows.sort(reverse=True)
    for d, n, m in rows[:6]:
        print(f'   {d:.3e}  (tensor max {m:.3e})  {n}')
    print('   -- along the backward path (relative to each tensor)')
    by = {n: (d, m) for d, n, m in rows}
    order = ['bbox_embed.layers.2.weight', 'bbox_embed.layers.0.weight'] + [f'transformer.decoder.layers.{i}.{t}' for i in (5, 4, 3, 2, 1, 0) for t in ('norm3.weight', 'linear2.weight', 'linear1.weight', 'norm2.weight', 'multihead_attn.out_proj.weight', 'multihead_attn.in_proj_weight', 'norm1.weight', 'self_attn.in_proj_weight')] + \
            [f'transformer.encoder.layers.{i}.{t}' for i in (5, 0) for t in ('norm2.weight', 'linear1.weight', 'self_attn.in_proj_weight')] + ['input_proj.weight', 'fc_in.layers.2.weight', 'fc_in.layers.0.weight']
    for n in order:
        if n in by:
            print(f'   {by[n][0] / (by[n][1] + 1e-30):.2e}  {n}')
