"""Plane-format ("P3") convolutions for the ResNet-50 trunk (csrc/p3_engine.hip; torchvision resnet50 behind
training/detr_backbone.py:98-114 in the reference).

An fp32 activation [N, H, W, C] travels between the trunk's kernels as its exact three-way bf16 split, a bfloat16 tensor of shape
[N, H, W, C/8, 3, 8] (hi / mid / lo planes of 8 channels side by side, 6 bytes per element): producers write it once in their epilogue,
consumers DMA it straight into LDS and contract on the bf16 matrix pipe with fp32-equivalent results.  Gradients of such tensors use the
same format, so a P3 tensor must have exactly ONE autograd consumer (planes cannot be summed by autograd's accumulation) — the trunk is
wired that way (`passthru`, see hip/conv.py::_ConvFn for the same hand-off on the fp32 path).
"""
import ctypes

import torch

from . import core
from .core import ACT_RELU
from .linear import act_backward
from .. import _lib


def p3_empty(N, H, W, C, device):
    return torch.empty((N, H, W, C // 8, 3, 8), device=device, dtype=torch.bfloat16)


def p3_dims(x):
    N, H, W, G, _, _ = x.shape
    return N, H, W, G * 8


def is_p3(x):
    return x is not None and x.dtype == torch.bfloat16 and x.dim() == 6


def split_raw(x):
    """fp32 [N, H, W, C] -> P3 (no autograd)."""
    x = core.f32c(x)
    N, H, W, C = x.shape
    out = p3_empty(N, H, W, C, x.device)
    core.check(core.lib().ldetr_p3_split_f32(core.ptr(x), C, core.ptr(out), N * H * W, C, core.stream()), 'p3_split')
    return out


def merge_raw(p):
    """P3 -> fp32 [N, H, W, C] (exact)."""
    N, H, W, C = p3_dims(p)
    out = torch.empty((N, H, W, C), device=p.device, dtype=torch.float32)
    core.check(core.lib().ldetr_p3_merge_f32(core.ptr(p), core.ptr(out), C, N * H * W, C, core.stream()), 'p3_merge')
    return out


class _SplitFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, pre):
        core.require_gpu(x)
        return pre if pre is not None else split_raw(x)

    @staticmethod
    def backward(ctx, dy):
        return merge_raw(dy.contiguous()), None


def split(x, pre=None):
    """fp32 [N, H, W, C] -> P3 with autograd (the gradient comes back merged).  pre: the split already computed by a raw pass (dual_trunk_forward)."""
    return _SplitFn.apply(x, pre)


class WeightPlanes(object):
    """P3 images of a list of conv weights, refreshed by ONE launch: [O][T][I] as stored (forward operand) and [I][T][O] with FrozenBN's
    scale folded in (data-gradient operand).  `convs`: list of (weight Parameter [O, I, KH, KW] in channels_last memory, bn module)."""

    def __init__(self, convs):
        self.convs = convs
        self.key = None
        self.table = None
        self.fwd = self.bwd = None
        self.blocks = 0
        self.offsets = []
        self.managed = False   # True: the training loop refreshes after every optimiser step (mark_stale + ensure); else every forward refreshes
        self.stale = True
        self.versions = None   # Tensor._version of every weight / folded scale at the last refresh: torch-level in-place writes (load_state_dict,
        #                        copy_params_and_buffers, broadcast, a torch optimiser, copy_) are detected here; only the raw flat-buffer Adam kernel,
        #                        which does not bump versions, needs the explicit refresh (training_loop.refresh_weight_planes)

    def _build(self, key):
        dev = self.convs[0][0].device
        total, offs = 0, []
        for w, _ in self.convs:
            offs.append(total)
            total += w.numel() * 6
        self.offsets = offs
        if self.fwd is None or self.fwd.numel() != total or self.fwd.device != dev:
            self.fwd = torch.empty(total, dtype=torch.uint8, device=dev)
            self.bwd = torch.empty(total, dtype=torch.uint8, device=dev)
        rows, blk = [], 0
        for (w, bn), off in zip(self.convs, offs):
            O, I, KH, KW = w.shape
            if not w.permute(0, 2, 3, 1).is_contiguous() or O % 8 or I % 8:
                raise RuntimeError('p3: conv weights must be channels_last with channel counts that are multiples of 8')
            sc = bn.folded()[0] if bn is not None else None
            rows.append([w.data_ptr(), sc.data_ptr() if sc is not None else 0, self.fwd.data_ptr() + off, self.bwd.data_ptr() + off, O, KH * KW, I, blk])
            blk += (w.numel() // 8 + 255) // 256
        self.table = torch.tensor(rows, dtype=torch.int64, device=dev)
        self.blocks = blk
        self.key = key

    def _key(self):
        return tuple((w.data_ptr(), bn.folded()[0].data_ptr() if bn is not None else 0) for w, bn in self.convs)

    def _versions(self):
        return tuple((w._version, bn.folded()[0]._version if bn is not None else 0) for w, bn in self.convs)

    def ensure(self):
        key = self._key()
        if key != self.key:
            self._build(key)
            self.stale = True
        vers = self._versions()
        if vers != self.versions:
            self.stale = True
        if self.stale or not self.managed:
            core.check(core.lib().ldetr_p3_weight_prep(core.ptr(self.table), len(self.convs), self.blocks, core.stream()), 'p3_weight_prep')
            self.stale = False
            self.versions = vers

    def ptrs(self, idx):
        off = self.offsets[idx]
        return self.fwd.data_ptr() + off, self.bwd.data_ptr() + off


def _epi(scale=None, shift=None, residual_p3=None, residual_f32=None, mask_p3=None, relu=False):
    ep = _lib.P3Epilogue()
    ep.alpha = 1.0
    ep.col_scale = scale.data_ptr() if scale is not None else None
    ep.col_bias = shift.data_ptr() if shift is not None else None
    ep.residual_p3 = residual_p3.data_ptr() if residual_p3 is not None else None
    ep.residual_f32 = residual_f32.data_ptr() if residual_f32 is not None else None
    ep.relu_mask_p3 = mask_p3.data_ptr() if mask_p3 is not None else None
    ep.relu = 1 if relu else 0
    return ep


# Replay: while set (a list of precomputed outputs, in call order), _ConvP3Fn.forward takes its result from the list instead of launching --
# the grouped forward of two trunks (conv_fwd_dual_raw, training/detr_backbone.py::dual_trunk_forward) computes both networks' activations in
# shared launches first; each network's ordinary forward then only builds its autograd graph around them.
_REPLAY = [None]


def conv_fwd_dual_raw(x1, x2, wf1, wf2, wshape, ep1, ep2, stride, pad, out_f32):
    """ldetr_p3_conv2d_fwd_dual on two P3 inputs of the same shape -> (y1, y2); no autograd.  wf*: device addresses of the forward weight images."""
    N, H, W, I = p3_dims(x1)
    O, _, KH, KW = wshape
    OH = (H + 2 * pad - KH) // stride + 1
    OW = (W + 2 * pad - KW) // stride + 1
    if out_f32:
        y1 = torch.empty((N, OH, OW, O), device=x1.device, dtype=torch.float32); y2 = torch.empty_like(y1)
        p1 = p2 = None; f1, f2 = y1, y2
    else:
        y1 = p3_empty(N, OH, OW, O, x1.device); y2 = p3_empty(N, OH, OW, O, x1.device)
        p1, p2 = y1, y2; f1 = f2 = None
    core.engine_call('ldetr_p3_conv2d_fwd', 4.0 * N * OH * OW * O * KH * KW * I, lambda: core.check(core.lib().ldetr_p3_conv2d_fwd_dual(
        core.ptr(x1), core.ptr(x2), N, H, W, I, ctypes.c_void_p(wf1), ctypes.c_void_p(wf2), O, KH, KW, stride, pad, ctypes.byref(ep1), ctypes.byref(ep2),
        core.ptr(p1), core.ptr(f1), core.ptr(p2), core.ptr(f2), core.stream()), 'p3_conv2d_fwd_dual'),
        nbytes=2 * (6.0 * (x1.numel() // 3 + O * I * KH * KW) + (4.0 if out_f32 else 6.0) * N * OH * OW * O))
    return y1, y2


class _ConvP3Fn(torch.autograd.Function):
    """y = act(conv(x, w) * scale + shift (+ residual)) on P3 activations; flags as hip/conv.py::_ConvFn.
    out_f32: the output leaves the P3 world (fp32 [N, OH, OW, O]); its incoming gradient is fp32 and the ReLU mask is applied here."""

    @staticmethod
    def forward(ctx, x, weight, wfwd, wbwd, scale, shift, residual, stride, pad, relu, premasked, mask_input, passthru, out_f32):
        core.require_gpu(x, weight)
        ctx.set_materialize_grads(False)
        N, H, W, I = p3_dims(x)
        O, _, KH, KW = weight.shape
        OH = (H + 2 * pad - KH) // stride + 1
        OW = (W + 2 * pad - KW) // stride + 1
        x = x.contiguous()
        res = residual.contiguous() if residual is not None else None
        ep = _epi(scale, shift, residual_p3=res, relu=relu)
        if _REPLAY[0] is not None:
            y = _REPLAY[0].pop(0)          # computed by the grouped launch of dual_trunk_forward
            if tuple(y.shape[:3]) != (N, OH, OW) or (y.dtype == torch.float32) != bool(out_f32):
                raise RuntimeError('p3 replay: precomputed activation does not match this convolution')
        elif out_f32:
            y = torch.empty((N, OH, OW, O), device=x.device, dtype=torch.float32)
            yp, yf = None, y
        else:
            y = p3_empty(N, OH, OW, O, x.device)
            yp, yf = y, None
        if _REPLAY[0] is None:
            core.engine_call('ldetr_p3_conv2d_fwd', 2.0 * N * OH * OW * O * KH * KW * I, lambda: core.check(core.lib().ldetr_p3_conv2d_fwd(
                core.ptr(x), N, H, W, I, ctypes.c_void_p(wfwd), O, KH, KW, stride, pad, ctypes.byref(ep), core.ptr(yp), core.ptr(yf), core.stream()), 'p3_conv2d_fwd'),
                nbytes=6.0 * (x.numel() // 3 + weight.numel() + (res.numel() // 3 if res is not None else 0)) + (4.0 if out_f32 else 6.0) * N * OH * OW * O)
        ctx.save_for_backward(x, scale, y if (relu and out_f32) else None)
        ctx.cfg = (stride, pad, relu, residual is not None, (N, H, W, I), (O, KH, KW, OH, OW), premasked, mask_input, out_f32, wbwd)
        ctx.wparam = weight
        if passthru:
            return y, x
        return y

    @staticmethod
    def backward(ctx, dy, dx_pass=None):
        x, sc, ysave = ctx.saved_tensors
        if dy is None:
            return (dx_pass,) + (None,) * 13
        stride, pad, relu, has_res, (N, H, W, I), (O, KH, KW, OH, OW), premasked, mask_input, out_f32, wbwd = ctx.cfg
        if out_f32:
            dy = core.f32c(dy)
            if relu:
                dy, _, _ = act_backward(dy.reshape(-1, O), ysave.reshape(-1, O), ACT_RELU, 0.0, 1.0, False)
            dyp = split_raw(dy.reshape(N, OH, OW, O))
        else:
            if relu and not premasked:
                raise RuntimeError('p3 conv: a ReLU output in P3 format must have its mask applied by its (single) consumer (premasked)')
            dyp = dy.contiguous()
        need_x = ctx.needs_input_grad[0]
        need_w = ctx.needs_input_grad[1] and not core.WEIGHT_GRADIENTS_DISABLED[0]
        need_res = has_res and ctx.needs_input_grad[6]
        dx = None
        flops = 2.0 * N * OH * OW * O * KH * KW * I
        if need_x and need_w:
            # both gradients in one launch (ldetr_p3_conv2d_bwd_pair): the weight gradient accumulates straight into the flat .grad when it can
            wparam = ctx.wparam
            gw = core.flat_grad(wparam)
            if gw is not None and gw.permute(0, 2, 3, 1).is_contiguous():
                dw_buf, dw = gw, None
            else:
                dw_buf = torch.zeros((O, I, KH, KW), device=x.device, dtype=torch.float32).contiguous(memory_format=torch.channels_last)
                dw = dw_buf
            dx = p3_empty(N, H, W, I, x.device)
            dxp = dx_pass.contiguous() if dx_pass is not None else None
            ep = _epi(residual_p3=dxp, mask_p3=x if mask_input else None)
            core.engine_call('ldetr_p3_conv2d_bwd_pair', 2.0 * flops, lambda: core.check(core.lib().ldetr_p3_conv2d_bwd_pair(
                core.ptr(dyp), N, OH, OW, O, ctypes.c_void_p(wbwd), core.ptr(x), I, KH, KW, stride, pad, H, W, ctypes.byref(ep), core.ptr(dx), None,
                core.ptr(sc), core.ptr(dw_buf), None, core.stream()), 'p3_conv2d_bwd_pair'),
                nbytes=6.0 * (dyp.numel() // 3 + wparam.numel() + 2 * N * H * W * I) + 6.0 * (x.numel() // 3) + 4.0 * wparam.numel())
            return (dx, dw, None, None, None, None, dyp if need_res else None) + (None,) * 7
        if need_x:
            dx = p3_empty(N, H, W, I, x.device)
            dxp = dx_pass.contiguous() if dx_pass is not None else None
            ep = _epi(residual_p3=dxp, mask_p3=x if mask_input else None)
            core.engine_call('ldetr_p3_conv2d_bwd_data', flops, lambda: core.check(core.lib().ldetr_p3_conv2d_bwd_data(
                core.ptr(dyp), N, OH, OW, O, ctypes.c_void_p(wbwd), I, KH, KW, stride, pad, H, W, ctypes.byref(ep), core.ptr(dx), None, core.stream()), 'p3_conv2d_bwd_data'),
                nbytes=6.0 * (dyp.numel() // 3 + ctx.wparam.numel() + 2 * N * H * W * I))
        elif dx_pass is not None:
            dx = dx_pass
        dw = None
        if need_w:
            wparam = ctx.wparam
            gw = core.flat_grad(wparam)
            if gw is not None and gw.permute(0, 2, 3, 1).is_contiguous():
                dw_buf, ret = gw, None
            else:
                dw_buf = torch.zeros((O, I, KH, KW), device=x.device, dtype=torch.float32).contiguous(memory_format=torch.channels_last)
                ret = dw_buf
            core.engine_call('ldetr_p3_conv2d_bwd_weight', flops, lambda: core.check(core.lib().ldetr_p3_conv2d_bwd_weight(
                core.ptr(x), N, H, W, I, core.ptr(dyp), O, KH, KW, stride, pad, core.ptr(sc), core.ptr(dw_buf), core.stream()), 'p3_conv2d_bwd_weight'),
                nbytes=6.0 * (x.numel() // 3 + dyp.numel() // 3) + 4.0 * wparam.numel())
            dw = ret
        dres = dyp if need_res else None
        return (dx, dw, None, None, None, None, dres) + (None,) * 7


def conv2d_p3(x, weight, wptrs, scale=None, shift=None, residual=None, stride=1, pad=0, relu=False, premasked=False, mask_input=False,
              passthru=False, out_f32=False):
    """wptrs = (forward image, data-gradient image) device addresses from WeightPlanes.ptrs()."""
    return _ConvP3Fn.apply(x, weight, wptrs[0], wptrs[1], scale, shift, residual, stride, pad, relu, premasked, mask_input, passthru, out_f32)
