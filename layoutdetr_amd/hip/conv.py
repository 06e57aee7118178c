"""NHWC implicit-GEMM convolutions for the ResNet-50 trunk and input_proj.

y = act( conv(x, w) * scale[c] + shift[c] (+ residual) ):  FrozenBatchNorm2d (a fixed affine,
training/detr_backbone.py:55-65), the bottleneck residual add and the ReLU are the conv epilogue, so
each conv block is one launch (the reference runs conv + 3-4 elementwise kernels).
Activations are [N, H, W, C] contiguous tensors; weights are the usual [O, I, KH, KW] parameters held
in channels_last memory (= OHWI contiguous) so state_dict keys/shapes stay torchvision-compatible.
"""
import ctypes

import torch

from . import core
from .core import ACT_NONE, ACT_RELU
from .linear import act_backward


def weight_ohwi(w):
    """[O, I, KH, KW] parameter -> contiguous [O, KH, KW, I] view (copy only if not channels_last)."""
    v = w.permute(0, 2, 3, 1)
    return v if v.is_contiguous() else v.contiguous()


def _grad_to_oihw(dw_ohwi):
    return dw_ohwi.permute(0, 3, 1, 2)  # logical OIHW, channels_last strides (matches the parameter layout)


class _ConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, scale, shift, residual, stride, pad, relu, x_is_nchw, premasked=False, mask_input=False, passthru=False):
        core.require_gpu(x, weight, scale, shift, residual)
        ctx.set_materialize_grads(False)
        w = core.f32c(weight_ohwi(weight))
        O, KH, KW, I = w.shape
        if x_is_nchw:
            xt = core.tensor4_nchw(x)
            N, H, W = x.shape[0], x.shape[2], x.shape[3]
        else:
            x = core.f32c(x)
            xt = core.tensor4_nhwc(x)
            N, H, W = x.shape[0], x.shape[1], x.shape[2]
        OH = (H + 2 * pad - KH) // stride + 1
        OW = (W + 2 * pad - KW) // stride + 1
        y = torch.empty((N, OH, OW, O), device=x.device, dtype=torch.float32)
        res = core.f32c(residual) if residual is not None else None
        sc = core.f32c(scale) if scale is not None else None
        sh = core.f32c(shift) if shift is not None else None
        ep = core.epilogue(col_scale=sc, col_bias=sh, residual=res, act=ACT_RELU if relu else ACT_NONE)
        core.engine_call('ldetr_conv2d_fwd_f32', 2.0 * N * OH * OW * O * KH * KW * I, lambda: core.check(core.lib().ldetr_conv2d_fwd_f32(
            core.ptr(x), ctypes.byref(xt), core.ptr(w), O, KH, KW, stride, pad, core.ptr(y), O, OH, OW, None, 0,
            ctypes.byref(ep), core.stream()), 'conv2d_fwd'), operands=(x, y, w, res))
        ctx.save_for_backward(x, w, sc, y if relu else None)
        ctx.cfg = (stride, pad, relu, x_is_nchw, residual is not None, shift is not None, (N, H, W, I), premasked, mask_input)
        ctx.params = (weight, shift)
        if passthru:
            # second output = x itself: the block's identity / downsample branch reads x through it, so this node is the only
            # autograd consumer of x and receives that branch's gradient (dx_pass) to add inside its own data-gradient epilogue
            return y, x
        return y

    @staticmethod
    def backward(ctx, dy, dx_pass=None):
        x, w, sc, y = ctx.saved_tensors
        if dy is None:
            return (dx_pass,) + (None,) * 11
        stride, pad, relu, x_is_nchw, has_res, has_shift, (N, H, W, I), premasked, mask_input = ctx.cfg
        O, KH, KW, _ = w.shape
        dy = core.f32c(dy)
        _, OH, OW, _ = dy.shape
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1] and not core.WEIGHT_GRADIENTS_DISABLED[0]
        need_shift = has_shift and ctx.needs_input_grad[3]
        need_res = has_res and ctx.needs_input_grad[4]
        dy2 = dy.reshape(-1, O)
        wparam, sparam = ctx.params
        gs = core.flat_grad(sparam) if (need_shift and O % 4 == 0) else None
        if relu and premasked and not need_shift:
            dpre2, dshift = dy2, None   # every consumer of this output already applied (y > 0) in its data-gradient epilogue
        elif relu:
            dpre2, dshift, _ = act_backward(dy2, y.reshape(-1, O), ACT_RELU, 0.0, 1.0, need_shift, dbias_out=gs)
        else:
            dpre2 = dy2
            dshift = None
            if need_shift:
                if gs is not None:
                    core.check(core.lib().ldetr_colsum_f32(core.ptr(dpre2), core.ptr(gs), 1, dpre2.shape[0], O, core.stream()), 'colsum')
                else:
                    dshift = core.colsum(dpre2).reshape(-1)
        dpre = dpre2.reshape(N, OH, OW, O)
        dyt = core.tensor4_nhwc(dpre)
        dx = dw = None
        if need_x:
            if x_is_nchw:
                raise RuntimeError('conv2d: gradient w.r.t. an NCHW image input is not implemented (never needed on the hot path)')
            dx = torch.empty((N, H, W, I), device=dy.device, dtype=torch.float32)
            # mask_input: x is the ReLU output of a producer whose only consumer is this conv -> its (x > 0) mask is applied here,
            # in the epilogue, and the producer skips its own activation-gradient pass (premasked)
            res2 = core.f32c(dx_pass).reshape(-1, I) if dx_pass is not None else None
            epb = None
            if mask_input or res2 is not None:
                epb = core.epilogue(residual=res2, mask_src=x.reshape(-1, I) if mask_input else None, mask_mode=1 if mask_input else 0)
            core.engine_call('ldetr_conv2d_bwd_data_f32', 2.0 * N * OH * OW * O * KH * KW * I, lambda: core.check(core.lib().ldetr_conv2d_bwd_data_f32(
                core.ptr(dpre), ctypes.byref(dyt), core.ptr(w), I, KH, KW, stride, pad, core.ptr(dx), I, H, W,
                core.ptr(sc), 0, ctypes.byref(epb) if epb is not None else None, core.stream()), 'conv2d_bwd_data'), operands=(dpre, dx, w))
        if need_w:
            gw = core.flat_grad(wparam)
            acc = 0
            if gw is not None and gw.permute(0, 2, 3, 1).is_contiguous():
                dw_ohwi, acc = gw.permute(0, 2, 3, 1), 1          # accumulate into the flat .grad view (OHWI memory)
            else:
                dw_ohwi = torch.empty((O, KH, KW, I), device=dy.device, dtype=torch.float32)
            xt = core.tensor4_nchw(x) if x_is_nchw else core.tensor4_nhwc(x)
            Kpix = N * OH * OW
            vec = (not x_is_nchw) and I % 4 == 0
            sk = 0   # the library picks tile and split-K factor together
            if sc is not None and not vec:
                # scalar-gather path has no operand scale: fold the BN scale into dy first
                dpre_s = dpre * sc
                dyt_s = core.tensor4_nhwc(dpre_s)
                core.engine_call('ldetr_conv2d_bwd_weight_f32', 2.0 * N * OH * OW * O * KH * KW * I, lambda: core.check(core.lib().ldetr_conv2d_bwd_weight_f32(
                    core.ptr(x), ctypes.byref(xt), core.ptr(dpre_s), ctypes.byref(dyt_s), core.ptr(dw_ohwi), KH, KW,
                    stride, pad, sk, None, 0, None, 0, acc, core.stream()), 'conv2d_bwd_weight'), operands=(x, dpre_s, dw_ohwi))
            else:
                def wgrad():
                    core.engine_call('ldetr_conv2d_bwd_weight_f32', 2.0 * N * OH * OW * O * KH * KW * I, lambda: core.check(core.lib().ldetr_conv2d_bwd_weight_f32(
                        core.ptr(x), ctypes.byref(xt), core.ptr(dpre), ctypes.byref(dyt), core.ptr(dw_ohwi), KH, KW, stride,
                        pad, sk, None, 0, core.ptr(sc), 0, acc, core.stream()), 'conv2d_bwd_weight'), operands=(x, dpre, dw_ohwi))
                wgrad()
            dw = None if acc else _grad_to_oihw(dw_ohwi)
        dres = dpre if need_res else None
        if dx is None and dx_pass is not None:
            dx = dx_pass
        return dx, dw, None, dshift, dres, None, None, None, None, None, None, None


def conv2d_nhwc(x, weight, scale=None, shift=None, residual=None, stride=1, pad=0, relu=False, x_is_nchw=False,
                premasked=False, mask_input=False, passthru=False):
    """premasked / mask_input: ReLU-gradient hand-off between a producer and its ONLY consumer (see _ConvFn.backward);
    set both ends together or neither.  passthru=True returns (y, x_alias): route every other use of x through x_alias and
    this node adds their gradient inside its data-gradient kernel (no separate add launch)."""
    return _ConvFn.apply(x, weight, scale, shift, residual, stride, pad, relu, x_is_nchw, premasked, mask_input, passthru)


class _MaxPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        core.require_gpu(x)
        x = core.f32c(x)
        N, H, W, C = x.shape
        OH, OW = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
        y = torch.empty((N, OH, OW, C), device=x.device, dtype=torch.float32)
        idx = torch.empty((N, OH, OW, C), device=x.device, dtype=torch.uint8)
        core.check(core.lib().ldetr_maxpool3x3s2_fwd_f32(core.ptr(x), core.ptr(y), core.ptr(idx), N, H, W, C,
                                                          core.stream()), 'maxpool_fwd')
        ctx.save_for_backward(idx)
        ctx.shape = (N, H, W, C)
        return y

    @staticmethod
    def backward(ctx, dy):
        idx, = ctx.saved_tensors
        N, H, W, C = ctx.shape
        dy = core.f32c(dy)
        dx = torch.empty((N, H, W, C), device=dy.device, dtype=torch.float32)
        core.check(core.lib().ldetr_maxpool3x3s2_bwd_f32(core.ptr(dy), core.ptr(idx), core.ptr(dx), N, H, W, C,
                                                          core.stream()), 'maxpool_bwd')
        return dx


def maxpool3x3s2_nhwc(x):
    return _MaxPoolFn.apply(x)


class _ConvTransposeFn(torch.autograd.Function):
    """y[n,oh,ow,co] = sum x[n,ih,iw,ci] * w[co,kh,kw,ci],  oh = ih*stride + kh - pad  (F.conv_transpose2d with the weight given
    un-transposed, OHWI): the plain (unmodulated) form behind torch_utils.ops.conv2d_gradfix.conv_transpose2d."""

    @staticmethod
    def forward(ctx, x, w_ohwi, stride, pad):
        core.require_gpu(x, w_ohwi)
        x = core.f32c(x); w = core.f32c(w_ohwi)
        N, H, W, I = x.shape
        O, KH, KW, _ = w.shape
        if I % 4 != 0:
            raise NotImplementedError('conv_transpose2d: input channel count must be a multiple of 4 on the gfx950 path')
        OH, OW = (H - 1) * stride - 2 * pad + KH, (W - 1) * stride - 2 * pad + KW
        y = torch.empty((N, OH, OW, O), device=x.device, dtype=torch.float32)
        xt = core.tensor4_nhwc(x)
        core.engine_call('ldetr_conv_transpose2d_fwd_f32', 2.0 * N * H * W * O * KH * KW * I, lambda: core.check(core.lib().ldetr_conv_transpose2d_fwd_f32(
            core.ptr(x), ctypes.byref(xt), core.ptr(w), O, KH, KW, stride, pad, core.ptr(y), O, OH, OW, None, 0, None, core.stream()), 'conv_transpose2d_fwd'), operands=(x, y, w))
        ctx.save_for_backward(x, w)
        ctx.cfg = (stride, pad)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        stride, pad = ctx.cfg
        N, H, W, I = x.shape
        O, KH, KW, _ = w.shape
        dy = core.f32c(dy)
        dyt = core.tensor4_nhwc(dy)
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            core.engine_call('ldetr_conv_transpose2d_bwd_data_f32', 2.0 * N * H * W * O * KH * KW * I, lambda: core.check(core.lib().ldetr_conv_transpose2d_bwd_data_f32(
                core.ptr(dy), ctypes.byref(dyt), core.ptr(w), I, KH, KW, stride, pad, core.ptr(dx), I, H, W, None, 0, None, core.stream()), 'conv_transpose2d_bwd_data'), operands=(dy, dx, w))
        if ctx.needs_input_grad[1] and not core.WEIGHT_GRADIENTS_DISABLED[0]:
            dw = torch.empty_like(w)
            xt = core.tensor4_nhwc(x)
            core.engine_call('ldetr_conv_transpose2d_bwd_weight_f32', 2.0 * N * H * W * O * KH * KW * I, lambda: core.check(core.lib().ldetr_conv_transpose2d_bwd_weight_f32(
                core.ptr(x), ctypes.byref(xt), core.ptr(dy), ctypes.byref(dyt), core.ptr(dw), KH, KW, stride, pad, 0, None, 0, None, 0, 0, core.stream()), 'conv_transpose2d_bwd_weight'), operands=(x, dy, dw))
        return dx, dw, None, None


def conv_transpose2d_nhwc(x, w_ohwi, stride=1, pad=0):
    return _ConvTransposeFn.apply(x, w_ohwi, stride, pad)
