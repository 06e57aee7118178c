"""Development aid: which forward view ops (slice / select / unbind / split / index ...) of grad-requiring tensors does one eager G+D iteration make, and
from where?  Each of them becomes a slice_backward / select_backward in the backward pass = one full-size zero fill + one copy (+ a fan-in add when a
tensor is sliced more than once).  Also lists the *_backward view ops actually executed.   python tools/trace_slices.py [per_gpu_batch]"""
import collections, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.utils._python_dispatch import TorchDispatchMode
import bench
from layoutdetr_amd.training import training_loop as tl
from layoutdetr_amd.training.loss import StyleGAN2Loss
from layoutdetr_amd.training.networks_detr import Discriminator, Generator

WATCH = ('aten.slice.Tensor', 'aten.select.int', 'aten.unbind', 'aten.split', 'aten.chunk', 'aten.narrow', 'aten.index', 'aten.expand', 'aten.cat', 'aten.stack', 'aten.permute', 'aten.transpose')


class Mode(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.fwd = collections.Counter(); self.bwd = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if 'backward' in name and any(v in name for v in ('slice', 'select', 'unbind', 'split', 'index', 'narrow', 'expand', 'embedding')):
            shape = 'x'.join(map(str, args[0].shape)) if args and isinstance(args[0], torch.Tensor) else ''
            self.bwd[(name, shape, str(args[1]) if len(args) > 1 and not isinstance(args[1], torch.Tensor) else '')] += 1
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
    # forward view ops: patch Tensor.__getitem__ & friends is intrusive; use the autograd graph instead: count grad_fn node types after each phase
    nodes = collections.Counter()
    orig_backward = torch.Tensor.backward

    def walk(t):
        seen, stack = set(), [t.grad_fn]
        while stack:
            fn = stack.pop()
            if fn is None or fn in seen:
                continue
            seen.add(fn)
            nm = type(fn).__name__
            if any(v in nm for v in ('Slice', 'Select', 'Unbind', 'Split', 'Index', 'Narrow', 'Expand', 'Cat', 'Stack', 'Copy', 'Clone', 'Add', 'Mul', 'Sum', 'Mean', 'View', 'Permute', 'Transpose', 'Squeeze', 'Unsqueeze')):
                meta = ''
                try:
                    meta = str(tuple(fn._saved_self_sym_sizes)) if hasattr(fn, '_saved_self_sym_sizes') else ''
                except Exception:
                    pass
                nodes[(nm, meta)] += 1
            stack.extend(f for f, _ in fn.next_functions)

    def spy_backward(self, *a, **k):
        walk(self)
        return orig_backward(self, *a, **k)
    torch.Tensor.backward = spy_backward
    try:
        with m:
            tl.training_iteration(loss, [pG, pD], dp, batch, b, z)
        torch.cuda.synchronize()
    finally:
        torch.Tensor.backward = orig_backward
    print('== autograd graph nodes of the view / glue kinds (both phases), by (type, input shape)')
    for (nm, meta), n in sorted(nodes.items(), key=lambda kv: -kv[1])[:80]:
        print(f'{n:5d}  {nm:32s} {meta}')
    print('== view-type backward ops executed')
    for (nm, shape, extra), n in sorted(m.bwd.items(), key=lambda kv: -kv[1])[:60]:
        print(f'{n:5d}  {nm:44s} grad {shape:18s} {extra}')


if __name__ == '__main__':
    main()
