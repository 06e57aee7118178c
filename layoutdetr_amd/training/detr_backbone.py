"""ResNet-50 trunk with frozen BatchNorm on the gfx950 implicit-GEMM conv kernels.

Interface of the reference `training/detr_backbone.py` (FrozenBatchNorm2d :29-65, BackboneBase :68-95,
Backbone :98-114, Joiner :117-134) with torchvision-0.13.1 ResNet-50 v1.5 parameter names
(`body.conv1.weight`, `body.layer2.0.downsample.0.weight`, ...), so UP-DETR / SwAV state dicts load.
Differences by design: activations are NHWC, each conv+BN(+residual)+ReLU is ONE kernel launch
(BN affine folded into the conv epilogue), and construction never touches the network (the
reference downloads SwAV weights at detr_backbone.py:110; load a local state_dict instead).
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F
from torch import nn

from ..detr_util.misc import NestedTensor
from ..hip import conv as hconv
from ..hip import p3 as hp3


class FrozenBatchNorm2d(nn.Module):
    """Fixed affine y = x * w*rsqrt(rv+eps) + (b - rm * w*rsqrt(rv+eps)); consumed as a conv epilogue."""

    def __init__(self, n):
        super().__init__()
        self.register_buffer('weight', torch.ones(n))
        self.register_buffer('bias', torch.zeros(n))
        self.register_buffer('running_mean', torch.zeros(n))
        self.register_buffer('running_var', torch.ones(n))
        self._folded = None

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        state_dict.pop(prefix + 'num_batches_tracked', None)
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)

    def folded(self):
        ver = (self.weight._version, self.bias._version, self.running_mean._version, self.running_var._version,
               self.weight.data_ptr())
        if self._folded is None or self._folded[0] != ver:
            scale = self.weight * (self.running_var + 1e-5).rsqrt()
            shift = self.bias - self.running_mean * scale
            self._folded = (ver, scale.contiguous(), shift.contiguous())
        return self._folded[1], self._folded[2]

    def forward(self, x):  # standalone use (NCHW), e.g. tests
        s, b = self.folded()
        return x * s.view(1, -1, 1, 1) + b.view(1, -1, 1, 1)

    def __deepcopy__(self, memo):
        new = FrozenBatchNorm2d(self.weight.numel())
        for k in ('weight', 'bias', 'running_mean', 'running_var'):
            getattr(new, k).data = getattr(self, k).data.clone()
        return new

    def __getstate__(self):
        d = dict(self.__dict__)
        d['_folded'] = None
        return d


class _Conv(nn.Module):
    """Conv2d parameters (OIHW shape, channels_last memory = OHWI contiguous), torchvision's kaiming fan_out init."""

    def __init__(self, cin, cout, k, stride=1, pad=0):
        super().__init__()
        w = torch.empty(cout, cin, k, k)
        nn.init.kaiming_normal_(w, mode='fan_out', nonlinearity='relu')
        self.weight = nn.Parameter(w.to(memory_format=torch.channels_last))
        self.stride, self.pad = stride, pad


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=False):
        super().__init__()
        self.conv1 = _Conv(inplanes, planes, 1)
        self.bn1 = FrozenBatchNorm2d(planes)
        self.conv2 = _Conv(planes, planes, 3, stride, 1)
        self.bn2 = FrozenBatchNorm2d(planes)
        self.conv3 = _Conv(planes, planes * 4, 1)
        self.bn3 = FrozenBatchNorm2d(planes * 4)
        self.downsample = None
        if downsample:
            self.downsample = nn.Sequential(_Conv(inplanes, planes * 4, 1, stride, 0), FrozenBatchNorm2d(planes * 4))

    # Set by ResNet50Body for blocks chained inside the trunk: `mask_in` = x is the ReLU output of the previous block, which
    # skips its own activation-gradient pass because this block applies (x > 0) for it; `premask_out` = the mirror flag.
    mask_in = False
    premask_out = False

    def forward(self, x):  # x: [N, H, W, C]
        s1, b1 = self.bn1.folded(); s2, b2 = self.bn2.folded(); s3, b3 = self.bn3.folded()
        # conv1 -> conv2 -> conv3 is a chain of single-consumer ReLU outputs: each consumer masks its data gradient with
        # (input > 0) in the GEMM epilogue, so conv1 and conv2 need no separate activation-gradient pass in backward
        # conv1 is the only autograd consumer of x: the identity / downsample branch reads x through conv1's pass-through alias, so
        # the gradient of x = conv1's data gradient + that branch's gradient (+ the previous block's ReLU mask) is ONE kernel
        out, x = hconv.conv2d_nhwc(x, self.conv1.weight, s1, b1, None, 1, 0, relu=True, premasked=True, mask_input=self.mask_in, passthru=True)
        out = hconv.conv2d_nhwc(out, self.conv2.weight, s2, b2, None, self.conv2.stride, 1, relu=True, premasked=True, mask_input=True)
        idt = x
        if self.downsample is not None:
            sd, bd = self.downsample[1].folded()
            idt = hconv.conv2d_nhwc(x, self.downsample[0].weight, sd, bd, None, self.downsample[0].stride, 0, relu=False)
        return hconv.conv2d_nhwc(out, self.conv3.weight, s3, b3, idt, 1, 0, relu=True, mask_input=True, premasked=self.premask_out)

    p3_index = None   # (conv1, conv2, conv3, downsample or None): this block's rows in ResNet50Body's hip.p3.WeightPlanes

    def forward_p3(self, x, planes, out_f32=False):
        """The same block on P3 activations (hip/p3.py): x and the result are plane-format tensors; out_f32: the block output leaves the
        trunk as fp32 [N, H, W, C] (its ReLU mask is then applied by this block's own backward)."""
        s1, b1 = self.bn1.folded(); s2, b2 = self.bn2.folded(); s3, b3 = self.bn3.folded()
        i1, i2, i3, idn = self.p3_index
        out, x = hp3.conv2d_p3(x, self.conv1.weight, planes.ptrs(i1), s1, b1, None, 1, 0, relu=True, premasked=True, mask_input=self.mask_in, passthru=True)
        out = hp3.conv2d_p3(out, self.conv2.weight, planes.ptrs(i2), s2, b2, None, self.conv2.stride, 1, relu=True, premasked=True, mask_input=True)
        idt = x
        if self.downsample is not None:
            sd, bd = self.downsample[1].folded()
            idt = hp3.conv2d_p3(x, self.downsample[0].weight, planes.ptrs(idn), sd, bd, None, self.downsample[0].stride, 0, relu=False)
        return hp3.conv2d_p3(out, self.conv3.weight, planes.ptrs(i3), s3, b3, idt, 1, 0, relu=True, mask_input=True,
                             premasked=self.premask_out and not out_f32, out_f32=out_f32)


class BackwardStages(object):
    """Cuts a phase's backward pass into stages so that the gradient exchange of the part that is already complete can run
    (RCCL, side stream) while the rest of the backward is still computing (reference site: training_loop.py:303-312 reduces
    after the whole backward; north_star asks for the overlap).

    Stage 1 = everything downstream of the trunk (heads, transformers, decoder) — runs inside `loss.backward()`;
    stage 2 = layer4 + layer3 of the trunk; stage 3 = layer2, layer1, stem.  `ResNet50Body.forward` calls `cut()` at the two
    boundaries: the activation is detached into a fresh leaf, so `loss.backward()` stops there and leaves the boundary gradient in
    `leaf.grad`; `run(stage)` then continues from it.  Parameter order in the flat gradient buffer is (stem, layer1..layer4, rest),
    so each stage completes one contiguous segment (`training_loop.FlatModule.stage_segments`)."""

    def __init__(self, n_stages=3):
        # n_stages = 2: trunk | rest only (no cut between layer2 and layer3).  At a few samples per GPU a stage is a few ms of kernels, and a
        # host that issues the collective between two graph replays a little late leaves the GPU idle for a visible part of it; two longer
        # stages hide that at the price of exchanging the whole trunk's gradient (94 MB) after the trunk's backward instead of in two pieces.
        self.n_stages = n_stages
        self.records = {2: [], 3: []}

    def cut(self, x, stage):
        if stage > self.n_stages:
            return x
        leaf = x.detach().requires_grad_(True)
        self.records[stage].append((x, leaf))
        return leaf

    def run(self, stage):
        recs, self.records[stage] = self.records[stage], []
        for x, leaf in recs:
            if leaf.grad is not None:
                x.backward(leaf.grad)

    def pending(self):
        return any(self.records.values())


import weakref
_P3_PLANES = weakref.WeakKeyDictionary()   # ResNet50Body -> hip.p3.WeightPlanes


def existing_p3_planes(body):
    """The body's weight images if it ever took the plane-format path, else None (nothing is built here: 2 x 6 bytes per trunk weight)."""
    return _P3_PLANES.get(body)


class ResNet50Body(nn.Module):
    """conv1/bn1/maxpool/layer1..4 of torchvision resnet50 (what IntermediateLayerGetter keeps, detr_backbone.py:78-79)."""

    stages = None   # a BackwardStages while a staged phase runs (set by training_loop.run_phase), else None

    def __init__(self):
        super().__init__()
        self.conv1 = _Conv(3, 64, 7, 2, 3)
        self.bn1 = FrozenBatchNorm2d(64)
        inplanes = 64
        for li, (planes, blocks) in enumerate(zip((64, 128, 256, 512), (3, 4, 6, 3)), start=1):
            stride = 1 if li == 1 else 2
            layers = [Bottleneck(inplanes, planes, stride, downsample=True)]
            inplanes = planes * 4
            layers += [Bottleneck(inplanes, planes) for _ in range(1, blocks)]
            setattr(self, f'layer{li}', nn.Sequential(*layers))
        blocks = [b for li in range(1, 5) for b in getattr(self, f'layer{li}')]
        for prev, nxt in zip(blocks[:-1], blocks[1:]):     # every block output feeds exactly the next block
            prev.premask_out = True
            nxt.mask_in = True

    def p3_planes(self):
        """P3 images of the 52 conv weights behind the stem (hip.p3.WeightPlanes; built on first use and kept OUTSIDE the module's state,
        so deepcopy / pickle / state_dict see the reference's module only)."""
        planes = _P3_PLANES.get(self)
        if planes is None:
            convs = []
            for li in range(1, 5):
                for b in getattr(self, f'layer{li}'):
                    i0 = len(convs)
                    convs += [(b.conv1.weight, b.bn1), (b.conv2.weight, b.bn2), (b.conv3.weight, b.bn3)]
                    idn = None
                    if b.downsample is not None:
                        idn = len(convs)
                        convs.append((b.downsample[0].weight, b.downsample[1]))
                    b.p3_index = (i0, i0 + 1, i0 + 2, idn)
            planes = _P3_PLANES[self] = hp3.WeightPlanes(convs)
        return planes

    @staticmethod
    def p3_enabled(x):
        """Plane-format trunk (default) unless LDETR_TRUNK_P3=0 or the activation would not fit the engine's 31-bit buffer offsets."""
        import os
        N, H, W, C = x.shape
        return os.environ.get('LDETR_TRUNK_P3', '1') != '0' and N * H * W * 256 * 6 < 0x7fffffff and N * H * W < (1 << 24)

    injected = None   # {(data_ptr, shape) of an input batch: its trunk output}, parked by dual_trunk_forward; consumed by the next forward on that batch

    def forward(self, x_nchw):
        if self.injected:
            hit = self.injected.pop((x_nchw.data_ptr(), tuple(x_nchw.shape)), None)
            if hit is not None:
                return hit
        x = self._entrance(x_nchw)
        if self.p3_enabled(x):
            return self._forward_p3(x)
        x = self.layer1(x); x = self.layer2(x)
        st = self.stages if (self.stages is not None and x.requires_grad and torch.is_grad_enabled()) else None
        if st is not None:
            x = st.cut(x, 3)
        x = self.layer3(x); x = self.layer4(x)
        if st is not None:
            x = st.cut(x, 2)
        return x  # [N, H/32, W/32, 2048]

    def _entrance(self, x_nchw):
        """Stem (7x7 / 2 conv + FrozenBN + ReLU on the fp32-operand engine) and max-pool -> fp32 [N, H/4, W/4, 64]."""
        s, b = self.bn1.folded()
        x = hconv.conv2d_nhwc(x_nchw, self.conv1.weight, s, b, None, 2, 3, relu=True, x_is_nchw=True)
        return hconv.maxpool3x3s2_nhwc(x)

    def _forward_p3(self, x, replay=None):
        """replay: the plane-format input followed by the 52 convolutions' precomputed outputs in call order (dual_trunk_forward); the pass then only
        builds the autograd graph."""
        planes = self.p3_planes()
        planes.ensure()
        x = hp3.split(x, pre=(replay.pop(0) if replay is not None else None))
        if replay is not None:
            prev, hp3._REPLAY[0] = hp3._REPLAY[0], replay
            try:
                y = self._blocks_p3(x, planes)
            finally:
                hp3._REPLAY[0] = prev
            if replay:
                raise RuntimeError('p3 replay: precomputed activations left over')
            return y
        return self._blocks_p3(x, planes)

    def _blocks_p3(self, x, planes):
        for b in self.layer1:
            x = b.forward_p3(x, planes)
        for b in self.layer2:
            x = b.forward_p3(x, planes)
        st = self.stages if (self.stages is not None and x.requires_grad and torch.is_grad_enabled()) else None
        if st is not None:
            x = st.cut(x, 3)
        for b in self.layer3:
            x = b.forward_p3(x, planes)
        n4 = len(self.layer4)
        for i, b in enumerate(self.layer4):
            x = b.forward_p3(x, planes, out_f32=(i == n4 - 1))
        if st is not None:
            x = st.cut(x, 2)
        return x  # fp32 [N, H/32, W/32, 2048]


def dual_trunk_forward(body_a, body_b, x_a, x_b):
    """body_a(x_a), body_b(x_b) for two ResNet50Body modules on same-shaped image batches, every plane-format convolution of the two trunks as ONE
    grouped launch (ldetr_p3_conv2d_fwd_dual).  G's and D's trunks convolve the same backgrounds through the same architecture with different
    weights (networks_detr.py:79-82 and :230-233 build one backbone each); each conv alone is a short launch whose blocks run in lock-step, and two
    of them in one grid overlap each other's fill and store burst.  Values are those of the two separate forwards bit for bit (same
    kernels, same tiles); each module's autograd graph is built by its own ordinary forward, replayed around the precomputed activations, so the
    two backward passes stay independent (they run in different phases)."""
    ea, eb = body_a._entrance(x_a), body_b._entrance(x_b)
    if not (body_a.p3_enabled(ea) and tuple(ea.shape) == tuple(eb.shape)):
        return body_a(x_a), body_b(x_b)
    planes_a, planes_b = body_a.p3_planes(), body_b.p3_planes()
    planes_a.ensure(); planes_b.ensure()
    outs_a, outs_b = [], []
    with torch.no_grad():
        ca, cb = hp3.split_raw(ea.detach()), hp3.split_raw(eb.detach())
        outs_a.append(ca); outs_b.append(cb)
        blocks_a = [b for li in range(1, 5) for b in getattr(body_a, f'layer{li}')]
        blocks_b = [b for li in range(1, 5) for b in getattr(body_b, f'layer{li}')]

        def dual(xa, xb, conv_a, conv_b, ia, ib, bn_a, bn_b, stride, pad, relu, res_a=None, res_b=None, out_f32=False):
            (sa, ha), (sb, hb) = bn_a.folded(), bn_b.folded()
            ya, yb = hp3.conv_fwd_dual_raw(xa, xb, planes_a.ptrs(ia)[0], planes_b.ptrs(ib)[0], conv_a.weight.shape, hp3._epi(sa, ha, residual_p3=res_a, relu=relu),
                                           hp3._epi(sb, hb, residual_p3=res_b, relu=relu), stride, pad, out_f32)
            outs_a.append(ya); outs_b.append(yb)
            return ya, yb
        for i, (ba, bb) in enumerate(zip(blocks_a, blocks_b)):
            (a1, a2, a3, ad), (b1, b2, b3, bd) = ba.p3_index, bb.p3_index
            o1a, o1b = dual(ca, cb, ba.conv1, bb.conv1, a1, b1, ba.bn1, bb.bn1, 1, 0, True)
            o2a, o2b = dual(o1a, o1b, ba.conv2, bb.conv2, a2, b2, ba.bn2, bb.bn2, ba.conv2.stride, 1, True)
            ida, idb = ca, cb
            if ba.downsample is not None:
                ida, idb = dual(ca, cb, ba.downsample[0], bb.downsample[0], ad, bd, ba.downsample[1], bb.downsample[1], ba.downsample[0].stride, 0, False)
            ca, cb = dual(o2a, o2b, ba.conv3, bb.conv3, a3, b3, ba.bn3, bb.bn3, 1, 0, True, ida, idb, out_f32=(i == len(blocks_a) - 1))
    return body_a._forward_p3(ea, replay=outs_a), body_b._forward_p3(eb, replay=outs_b)


class BackboneBase(nn.Module):
    def __init__(self, backbone: nn.Module, train_backbone: bool, num_channels: int, return_interm_layers: bool):
        super().__init__()
        for name, parameter in backbone.named_parameters():
            if not train_backbone or 'layer2' not in name and 'layer3' not in name and 'layer4' not in name:
                parameter.requires_grad_(False)
        if return_interm_layers:
            raise NotImplementedError('return_interm_layers is not used by LayoutDETR')
        self.body = backbone
        self.num_channels = num_channels

    def forward(self, tensor_list):
        if isinstance(tensor_list, NestedTensor):
            x = self.body(tensor_list.tensors).permute(0, 3, 1, 2)  # NCHW-shaped view of NHWC memory
            m = tensor_list.mask
            assert m is not None
            if getattr(tensor_list, 'uniform', False):
                mask = torch.zeros((x.shape[0], x.shape[2], x.shape[3]), dtype=torch.bool, device=x.device)
            else:
                mask = F.interpolate(m[None].float(), size=x.shape[-2:]).to(torch.bool)[0]
            return OrderedDict([('0', NestedTensor(x, mask, getattr(tensor_list, 'uniform', False)))])
        return OrderedDict([('0', self.body(tensor_list).permute(0, 3, 1, 2))])


class Backbone(BackboneBase):
    """ResNet-50 backbone with frozen BatchNorm.  `name` must be 'resnet50'."""

    def __init__(self, name: str = 'resnet50', train_backbone: bool = True, return_interm_layers: bool = False, dilation: bool = False):
        if name != 'resnet50' or dilation:
            raise NotImplementedError('only the resnet50 / no-dilation configuration of LayoutDETR is implemented')
        super().__init__(ResNet50Body(), train_backbone, 2048, return_interm_layers)


class Joiner(nn.Sequential):
    def __init__(self, backbone, position_embedding):
        super().__init__(backbone, position_embedding)

    def forward(self, tensor_list):
        if isinstance(tensor_list, NestedTensor):
            xs = self[0](tensor_list)
            out, pos = [], []
            for _, x in xs.items():
                out.append(x)
                pos.append(self[1](x).to(x.tensors.dtype))
            return out, pos
        return list(self[0](tensor_list).values())
