"""Count zero-fills / copies issued from Python call sites during one eager G+D iteration (development aid)."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
cnt = collections.Counter()
def where():
    f = sys._getframe(2)
    while f is not None and ('layoutdetr_amd' not in f.f_code.co_filename and 'bench.py' not in f.f_code.co_filename):
        f = f.f_back
    return 'torch-internal' if f is None else f'{f.f_code.co_filename.split("layoutdetr_amd/")[-1]}:{f.f_lineno}'
def wrap(obj, name, tag, cond=None):
    orig = getattr(obj, name)
    def w(*a, **k):
        if cond is None or cond(*a, **k):
            cnt[(tag, where())] += 1
        return orig(*a, **k)
    setattr(obj, name, w)
wrap(torch, 'zeros', 'zeros'); wrap(torch, 'zeros_like', 'zeros_like'); wrap(torch.Tensor, 'zero_', 'zero_'); wrap(torch.Tensor, 'new_zeros', 'new_zeros')
wrap(torch.Tensor, 'contiguous', 'contiguous(copy)', lambda t, *a, **k: not t.is_contiguous(**k))
wrap(torch.Tensor, 'clone', 'clone'); wrap(torch.Tensor, 'copy_', 'copy_'); wrap(torch, 'cat', 'cat'); wrap(torch.Tensor, 'float', 'float(copy)', lambda t: t.dtype != torch.float32)
wrap(torch.Tensor, 'to', 'to'); wrap(torch.Tensor, 'reshape', 'reshape(copy)', lambda t, *a: not t.is_contiguous())
wrap(torch.Tensor, 'sum', 'sum'); wrap(torch.Tensor, '__add__', 'add'); wrap(torch.Tensor, '__mul__', 'mul'); wrap(torch.Tensor, '__iadd__', 'iadd')
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
it(); torch.cuda.synchronize(); cnt.clear()
it(); torch.cuda.synchronize()
for (tag, w), n in cnt.most_common(80):
    print(f'{n:5d} {tag:18s} {w}')
