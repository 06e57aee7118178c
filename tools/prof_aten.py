"""Which Python lines issue the small ATen kernels (copies / adds / fills) of one eager G+D iteration (development aid)."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from bench import make_batch, to_device_batch
from layoutdetr_amd.training import training_loop as tl
from layoutdetr_amd.training.loss import StyleGAN2Loss
from layoutdetr_amd.training.networks_detr import Discriminator, Generator
dev = torch.device('cuda:0'); B = 16; bg = 256
torch.manual_seed(0)
kw = dict(num_bbox_labels=8, img_channels=3, img_height=bg, img_width=bg, c_dim=0, background_size=bg, bert_f_dim=768, im_f_dim=512)
G = Generator(z_dim=4, **kw).train().requires_grad_(False).to(dev); D = Discriminator(**kw).train().requires_grad_(False).to(dev)
G.static_shapes = True; D.static_shapes = True
pG = tl.Phase('Gmain', G, lr=1e-5); pD = tl.Phase('Dmain', D, lr=1e-5)
loss = StyleGAN2Loss(dev, G, D); dp = tl.DataParallelStep(1)
batch = to_device_batch(make_batch(B, bg, dev, 1), dev)
z = torch.randn(B, 9, 4, device=dev)
it = lambda: tl.training_iteration(loss, [pG, pD], dp, batch, B, [z, z])
it(); torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=False) as prof:
    it(); torch.cuda.synchronize()
agg = collections.Counter()
want = ('aten::copy_', 'aten::add', 'aten::add_', 'aten::zero_', 'aten::mul', 'aten::sum', 'aten::div', 'aten::fill_')
def top_parent(ev):
    names = []
    p = ev.cpu_parent
    while p is not None:
        names.append(p.name)
        p = p.cpu_parent
    for n in names:
        if 'Backward' in n or 'AccumulateGrad' in n:
            return n
    return names[-1] if names else 'top'
for ev in prof.events():
    if ev.name in want:
        par = ev.cpu_parent.name if ev.cpu_parent is not None else 'top'
        if ev.name == 'aten::fill_' and par == 'aten::zero_':
            continue
        agg[(ev.name, top_parent(ev))] += 1
for (name, where), n in agg.most_common(60):
    print(f'{n:5d}  {name:14s} {where}')
