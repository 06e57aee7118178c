"""Soak check (development aid): N hipGraph-replayed G+D iterations of the bench workload with dropout on, then every parameter, Adam moment
and G_ema value must be finite and the parameters must have moved.  Usage: python tools/soak_step.py [iterations] [per_gpu_batch]"""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from layoutdetr_amd.training import training_loop as tl
from layoutdetr_amd.training.loss import StyleGAN2Loss
from layoutdetr_amd.training.networks_detr import Discriminator, Generator

n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 60
b = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device('cuda:0')
torch.manual_seed(0)
kw = dict(num_bbox_labels=8, img_channels=3, img_height=256, img_width=256, c_dim=0, background_size=256, bert_f_dim=768, im_f_dim=512)
G = Generator(z_dim=4, **kw).train().requires_grad_(False).to(dev)
D = Discriminator(**kw).train().requires_grad_(False).to(dev)
G.static_shapes = D.static_shapes = True
G_ema = copy.deepcopy(G).eval()
pG, pD = tl.Phase('Gmain', G, lr=2e-4, betas=(0.0, 0.99), eps=1e-8), tl.Phase('Dmain', D, lr=2e-4, betas=(0.0, 0.99), eps=1e-8)
ema = tl.EmaTracker(pG, G_ema)
loss = StyleGAN2Loss(dev, G, D, share_D_trunk='iteration')
dp = tl.DataParallelStep(1)
batch = bench.to_device_batch(bench.make_batch(b, 256, dev, 1), dev)
p0 = [pG.fm.flat.clone(), pD.fm.flat.clone()]
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        tl.training_iteration(loss, [pG, pD], dp, batch, b, [torch.randn(b, 9, 4, device=dev) for _ in range(2)], ema=ema, batch_size=b, ema_kimg=b * 10 / 32, cur_nimg=0)
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
g = tl.GraphedIteration(loss, [pG, pD], dp, batch, b, 4, ema=ema, batch_size=b, ema_kimg=b * 10 / 32, capture_stream=side)
for i in range(n_it):
    g.run()
torch.cuda.synchronize()
ok = True
for name, ph, q0 in (('G', pG, p0[0]), ('D', pD, p0[1])):
    fin = bool(torch.isfinite(ph.fm.flat).all()) and bool(torch.isfinite(ph.fm.gflat).all())
    moved = float((ph.fm.flat - q0).abs().max())
    gmax = float(ph.fm.gflat.abs().max())
    print(f'{name}: parameters finite={fin}, max |delta p| = {moved:.3e}, last max |grad| = {gmax:.3e}')
    ok = ok and fin and moved > 0
fe = all(bool(torch.isfinite(p).all()) for p in G_ema.parameters())
print('G_ema finite:', fe)
print('SOAK', 'OK' if (ok and fe) else 'FAILED', f'({n_it} replayed iterations, {b} samples)')
