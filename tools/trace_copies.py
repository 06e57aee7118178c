"""Development aid: which Python call sites materialise tensor copies (contiguous / clone / copy_ / to) during one eager iteration."""
import collections
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
from layoutdetr_amd.training import training_loop as tl
from layoutdetr_amd.training.loss import StyleGAN2Loss
from layoutdetr_amd.training.networks_detr import Discriminator, Generator

counts = collections.Counter()
ON = [False]


def where():
    for fr in reversed(traceback.extract_stack(limit=30)[:-2]):
        if 'layoutdetr_amd' in fr.filename:
            return f'{os.path.relpath(fr.filename, ROOT)}:{fr.lineno}'
    return 'other'


def wrap(name):
    orig = getattr(torch.Tensor, name)

    def f(self, *a, **k):
        r = orig(self, *a, **k)
        if ON[0] and isinstance(r, torch.Tensor) and self.is_cuda and (name == 'copy_' or r.data_ptr() != self.data_ptr()):
            counts[(name, where(), 'x'.join(map(str, self.shape)), 'contig' if self.is_contiguous() else 'strided')] += 1
        return r
    setattr(torch.Tensor, name, f)


for n in ('contiguous', 'clone', 'copy_', 'to', 'float'):
    wrap(n)


def main():
    b = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    kw = dict(num_bbox_labels=8, img_channels=3, img_height=256, img_width=256, c_dim=0, background_size=256, bert_f_dim=768, im_f_dim=512)
    G = Generator(z_dim=4, **kw).train().requires_grad_(False).to(dev)
    D = Discriminator(**kw).train().requires_grad_(False).to(dev)
    G.static_shapes = D.static_shapes = True
    pG, pD = tl.Phase('Gmain', G, lr=1e-5), tl.Phase('Dmain', D, lr=1e-5)
    loss = StyleGAN2Loss(dev, G, D, share_D_trunk='iteration')
    dp = tl.DataParallelStep(1)
    batch = bench.to_device_batch(bench.make_batch(b, 256, dev, 1), dev)
    z = [torch.randn(b, 9, 4, device=dev) for _ in range(2)]
    tl.training_iteration(loss, [pG, pD], dp, batch, b, z)
    torch.cuda.synchronize()
    ON[0] = True
    tl.training_iteration(loss, [pG, pD], dp, batch, b, z)
    torch.cuda.synchronize()
    ON[0] = False
    print(sum(counts.values()), 'python-level copies in one iteration')
    for (name, w, shape, c), n in counts.most_common(60):
        print(f'{n:5d} {name:10s} {w:58s} {shape:20s} {c}')


if __name__ == '__main__':
    main()
