"""Development aid: which ATen ops (i.e. non-ldetr launches) does one eager G+D iteration issue, and from where?
python tools/trace_aten.py [per_gpu_batch]  -> table of (op, python source line | autograd) counts."""
import collections
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.utils._python_dispatch import TorchDispatchMode

import bench
from layoutdetr_amd.training import training_loop as tl
from layoutdetr_amd.training.loss import StyleGAN2Loss
from layoutdetr_amd.training.networks_detr import Discriminator, Generator

VIEW = ('view', 'reshape', 'permute', 'transpose', 'expand', 'slice', 'select', 'unsqueeze', 'squeeze', 'detach', 'alias', 'as_strided', 't.default',
        'unbind', 'split', 'size', 'stride', 'is_', '_unsafe_view', 'narrow', 'flatten', 'empty', 'sym_', 'numel', 'dim', 'lift_fresh', '_local_scalar', 'set_',
        'record_stream', 'result_type', 'can_cast', 'item', 'unflatten', 'chunk', 'view_as', 'new_empty', 'resize_')


class Mode(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.counts = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(v in name for v in VIEW):
            where = 'autograd/C++'
            for fr in reversed(traceback.extract_stack(limit=40)):
                if 'layoutdetr_amd' in fr.filename and 'trace_aten' not in fr.filename:
                    where = f'{os.path.relpath(fr.filename, ROOT)}:{fr.lineno}'
                    break
            shape = ''
            for a in args:
                if isinstance(a, torch.Tensor):
                    shape = 'x'.join(map(str, a.shape)); break
            self.counts[(name, where, shape)] += 1
        return func(*args, **(kwargs or {}))


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
    m = Mode()
    with m:
        tl.training_iteration(loss, [pG, pD], dp, batch, b, z)
    torch.cuda.synchronize()
    tot = sum(m.counts.values())
    print(f'{tot} non-view ATen calls in one iteration')
    agg = collections.Counter()
    for (name, where, shape), n in m.counts.items():
        agg[(name, where)] += n
    for (name, where), n in agg.most_common(140):
        shapes = [f"{s}:{c}" for (nm, w, s), c in m.counts.items() if nm == name and w == where][:12]
        print(f'{n:5d}  {name:42s} {where:60s} {" ".join(shapes)}')


if __name__ == '__main__':
    main()
