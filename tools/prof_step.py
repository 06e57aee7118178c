"""Timing breakdown of the G+D step (development aid)."""
import sys, time, copy; sys.path.insert(0, '.'); sys.path.insert(0, '..')
import torch
from bench import make_batch, to_device_batch
from layoutdetr_amd.training import training_loop as tl
from layoutdetr_amd.training.loss import StyleGAN2Loss
from layoutdetr_amd.training.networks_detr import Discriminator, Generator
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
bg = int(sys.argv[2]) if len(sys.argv) > 2 else 256
torch.manual_seed(0)
kw = dict(num_bbox_labels=8, img_channels=3, img_height=bg, img_width=bg, c_dim=0, background_size=bg, bert_f_dim=768, im_f_dim=512)
G = Generator(z_dim=4, **kw).train().requires_grad_(False).to(dev); D = Discriminator(**kw).train().requires_grad_(False).to(dev)
pG = tl.Phase('Gmain', G, lr=1e-5); pD = tl.Phase('Dmain', D, lr=1e-5)
loss = StyleGAN2Loss(dev, G, D); dp = tl.DataParallelStep(1)
batch = to_device_batch(make_batch(B, bg, dev, 1), dev)
def T(name, fn, n=2):
    fn(); torch.cuda.synchronize(); t = time.time()
    for _ in range(n): r = fn()
    torch.cuda.synchronize(); print(f'{name}: {(time.time()-t)/n*1e3:.1f} ms', flush=True); return r
z = torch.randn(B, 9, 4, device=dev)
a = (batch['bbox_class'], batch['bbox_real'], batch['bbox_text'], batch['bbox_patch'], batch['padding_mask'], batch['background'], None)
with torch.no_grad():
    T('G backbone fwd', lambda: G.backbone.__getitem__(0).body(batch['background']))
    T('G fwd (no grad)', lambda: G(z, *a))
    T('D fwd (no grad)', lambda: D(batch['bbox_real'], batch['bbox_class'], batch['bbox_text'], batch['bbox_patch'], batch['padding_mask'], batch['background'], None))
    x0 = torch.randn(B, 256, device=dev)
    T('D.bg_decoder fwd (no grad)', lambda: D.bg_decoder(x0))
G.requires_grad_(True)
def gfb():
    G.zero_grad(set_to_none=False); out = G(z, *a, reconst=True); (out[0].sum() + out[1] + out[4]).backward()
T('G fwd+bwd', gfb)
def bbfb():
    y = G.backbone[0].body(batch['background']); y.sum().backward()
T('G backbone fwd+bwd', bbfb)
G.requires_grad_(False); D.requires_grad_(True)
def decfb():
    img = D.bg_decoder(x0); img.square().mean().backward()
T('D.bg_decoder fwd+bwd', decfb)
D.requires_grad_(False)
T('full iteration', lambda: tl.training_iteration(loss, [pG, pD], dp, batch, B, [z, z]))
