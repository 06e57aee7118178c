"""StyleGAN2 modulated convolution layers as fused gfx950 launches (forward + hand-written backward).

Reference chain (non-fused branch, training/networks_stylegan2.py:57-75 + SynthesisLayer :307-326):
    x*styles -> conv2d_resample (conv | transposed conv + 4x4 FIR) -> *dcoefs -> bias_act(lrelu, sqrt2)
i.e. one conv plus 4-5 full-tensor elementwise passes per layer.  Here:
    3x3 layer : ONE implicit-GEMM launch (style scale in the A-operand loader; dcoefs, bias, lrelu in the epilogue)
    up layer  : transposed-conv launch (style scale in loader, dcoefs in epilogue) + ONE FIR launch with
                bias + lrelu fused (demodulation commutes with the per-channel FIR)
    toRGB     : one launch (style scale in loader, bias in epilogue)
Activations are NHWC [B, H, W, C]; weights are [O, I, k, k] parameters in channels_last memory (OHWI).
"""
import ctypes
import math

import torch

from . import core
from .conv import weight_ohwi, _grad_to_oihw
from .core import ACT_LRELU, ACT_NONE
from .linear import act_backward
from ..torch_utils.ops import upfirdn2d as _up


def _mul_reduce(a, x, scale, B, P, C, want_out=True):
    out = torch.empty_like(a) if want_out else None
    red = core.zeros((B, C), a.device)
    core.check(core.lib().ldetr_mul_reduce_f32(core.ptr(a), core.ptr(x), core.ptr(scale), core.ptr(out), core.ptr(red),
                                               B, P, C, core.stream()), 'mul_reduce')
    return out, red


def _prescaled_for_wgrad(x, dv, s, d, B, H):
    """Operands of a layer's weight gradient dW = sum_b s_b (x_b (*) (dv_b d_b)): -> (x, dv, style scale, demodulation scale) to hand to the weight-gradient
    launch.  The launch can apply the per-sample scales itself (K slices aligned to samples, partial tiles merged by atomics): right for the large
    layers.  On the 4x4 .. 16x16 layers at >= 8 samples that costs more than it saves -- 16 slices of 16 .. 256 pixels each write a full
    [Cout, 9 Cin] partial (150 MB of atomics at 512 channels; measured at 16 samples: 8x8 162 -> 66 us, 4x4 58 -> 42 us, 16x16 266 -> 211 us,
    tools/bench_sg2_wgrad.py) -- so the two small operands are scaled first (two element-wise launches over <= 8 MB) and the reduction is split
    freely.  At 2 samples per GPU or from 32x32 on the in-launch scales win."""
    if B >= 8 and H <= 16:
        return x * s[:, None, None, :], dv * d[:, None, None, :], None, None
    return x, dv, s, d


class _ModConvFn(torch.autograd.Function):
    """y = lrelu(conv3x3(x * s) * d + bias) * gain      (up == 1)"""

    @staticmethod
    def forward(ctx, x, weight, styles, dcoefs, bias, pad, act_alpha, act_gain):
        core.require_gpu(x, weight, styles, dcoefs, bias)
        x = core.f32c(x); w = core.f32c(weight_ohwi(weight)); s = core.f32c(styles); d = core.f32c(dcoefs); b = core.f32c(bias)
        B, H, W, I = x.shape
        O, KH, KW, _ = w.shape
        OH, OW = H + 2 * pad - KH + 1, W + 2 * pad - KW + 1
        y = torch.empty((B, OH, OW, O), device=x.device, dtype=torch.float32)
        xt = core.tensor4_nhwc(x)
        ep = core.epilogue(samp_scale=d, col_bias=b, act=ACT_LRELU, act_alpha=act_alpha, act_gain=act_gain)
        core.engine_call('ldetr_conv2d_fwd_f32', 2.0 * B * OH * OW * O * KH * KW * I, lambda: core.check(core.lib().ldetr_conv2d_fwd_f32(core.ptr(x), ctypes.byref(xt), core.ptr(w), O, KH, KW, 1, pad, core.ptr(y), O,
                                                   OH, OW, core.ptr(s), s.stride(0), ctypes.byref(ep), core.stream()), 'modconv_fwd'), operands=(x, y, w))
        ctx.save_for_backward(x, w, s, d, b, y)
        ctx.cfg = (pad, act_alpha, act_gain)
        ctx.params = (weight, bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, s, d, b, y = ctx.saved_tensors
        pad, act_alpha, act_gain = ctx.cfg
        B, H, W, I = x.shape
        O, KH, KW, _ = w.shape
        _, OH, OW, _ = y.shape
        dy = core.f32c(dy)
        wparam, bparam = ctx.params
        dv2, dbias, ddemod = act_backward(dy.reshape(-1, O), y.reshape(-1, O), ACT_LRELU, act_alpha, act_gain, True,
                                          bias=b, demod=d, want_ddemod=True, B=B, dbias_out=core.flat_grad(bparam))
        dv = dv2.reshape(B, OH, OW, O)
        dvt = core.tensor4_nhwc(dv)
        dx = dw = ds = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[2]:
            dxs = torch.empty((B, H, W, I), device=dy.device, dtype=torch.float32)
            core.engine_call('ldetr_conv2d_bwd_data_f32', 2.0 * B * OH * OW * O * KH * KW * I, lambda: core.check(core.lib().ldetr_conv2d_bwd_data_f32(core.ptr(dv), ctypes.byref(dvt), core.ptr(w), I, KH, KW, 1, pad,
                                                            core.ptr(dxs), I, H, W, core.ptr(d), d.stride(0), None, core.stream()),
                       'modconv_bwd_data'), operands=(dv, dxs, w))
            dx, ds = _mul_reduce(dxs, x, s, B, H * W, I)
        if ctx.needs_input_grad[1]:
            gw = core.flat_grad(wparam)
            acc = 0
            if gw is not None and gw.permute(0, 2, 3, 1).is_contiguous():
                dw_ohwi, acc = gw.permute(0, 2, 3, 1), 1
            else:
                dw_ohwi = torch.empty((O, KH, KW, I), device=dy.device, dtype=torch.float32)
            xt = core.tensor4_nhwc(x)
            sk = 0   # the library picks tile and split-K factor together
            xw, dvw, sw, dw_ = _prescaled_for_wgrad(x, dv, s, d, B, H)
            core.engine_call('ldetr_conv2d_bwd_weight_f32', 2.0 * B * OH * OW * O * KH * KW * I, lambda: core.check(core.lib().ldetr_conv2d_bwd_weight_f32(core.ptr(xw), ctypes.byref(xt), core.ptr(dvw), ctypes.byref(dvt),
                                                              core.ptr(dw_ohwi), KH, KW, 1, pad, sk, core.ptr(sw), sw.stride(0) if sw is not None else 0,
                                                              core.ptr(dw_), dw_.stride(0) if dw_ is not None else 0, acc, core.stream()), 'modconv_bwd_weight'), operands=(x, dv, dw_ohwi))
            dw = None if acc else _grad_to_oihw(dw_ohwi)
        return dx, dw, ds, ddemod, dbias, None, None, None


class _ModConvUpFn(torch.autograd.Function):
    """y = lrelu(FIR4x4(conv_transpose3x3_s2(x * s) * d) * 4 + bias) * gain      (up == 2)"""

    @staticmethod
    def forward(ctx, x, weight, styles, dcoefs, bias, f, act_alpha, act_gain):
        core.require_gpu(x, weight, styles, dcoefs, bias, f)
        x = core.f32c(x); w = core.f32c(weight_ohwi(weight)); s = core.f32c(styles); d = core.f32c(dcoefs); b = core.f32c(bias)
        B, H, W, I = x.shape
        O, KH, KW, _ = w.shape
        UH, UW = (H - 1) * 2 + KH, (W - 1) * 2 + KW
        ud = torch.empty((B, UH, UW, O), device=x.device, dtype=torch.float32)
        xt = core.tensor4_nhwc(x)
        ep = core.epilogue(samp_scale=d)
        core.engine_call('ldetr_conv_transpose2d_fwd_f32', 2.0 * B * H * W * O * KH * KW * I, lambda: core.check(core.lib().ldetr_conv_transpose2d_fwd_f32(core.ptr(x), ctypes.byref(xt), core.ptr(w), O, KH, KW, 2, 0, core.ptr(ud),
                                                             O, UH, UW, core.ptr(s), s.stride(0), ctypes.byref(ep), core.stream()),
                   'modconv_up_fwd'), operands=(x, ud, w))
        # conv2d_resample.py:113-130 with padding=1, 4-tap filter, up=2 -> transposed-conv pad 0, FIR pad [1,1,1,1], gain 4
        y = _up._kernel_call(ud.permute(0, 3, 1, 2), f, 1, 1, 1, 1, 1, 1, 1, 1, False, 4.0, act_bias=b,
                             act=(act_alpha, act_gain)).permute(0, 2, 3, 1)
        ctx.save_for_backward(x, w, s, d, b, f, ud, y)
        ctx.cfg = (act_alpha, act_gain)
        ctx.params = (weight, bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, s, d, b, f, ud, y = ctx.saved_tensors
        act_alpha, act_gain = ctx.cfg
        B, H, W, I = x.shape
        O, KH, KW, _ = w.shape
        _, UH, UW, _ = ud.shape
        dy = core.f32c(dy)
        wparam, bparam = ctx.params
        dv2, dbias, _ = act_backward(dy.reshape(-1, O), y.reshape(-1, O), ACT_LRELU, act_alpha, act_gain, True,
                                     dbias_out=core.flat_grad(bparam))
        dv = dv2.reshape(y.shape)
        # adjoint of the FIR (upfirdn2d.py:252-270): flipped taps, pad [fw-px0-1, iw-ow+px0, ...] = [2,2,2,2]
        dud = _up._kernel_call(dv.permute(0, 3, 1, 2), f, 1, 1, 1, 1, 2, 2, 2, 2, True, 4.0).permute(0, 2, 3, 1)
        _, dd_num = _mul_reduce(dud, ud, None, B, UH * UW, O, want_out=False)
        ddemod = dd_num / d
        dudt = core.tensor4_nhwc(dud)
        dx = dw = ds = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[2]:
            dxs = torch.empty((B, H, W, I), device=dy.device, dtype=torch.float32)
            core.engine_call('ldetr_conv_transpose2d_bwd_data_f32', 2.0 * B * H * W * O * KH * KW * I, lambda: core.check(core.lib().ldetr_conv_transpose2d_bwd_data_f32(core.ptr(dud), ctypes.byref(dudt), core.ptr(w), I, KH, KW, 2, 0,
                                                                      core.ptr(dxs), I, H, W, core.ptr(d), d.stride(0), None,
                                                                      core.stream()), 'modconv_up_bwd_data'), operands=(dud, dxs, w))
            dx, ds = _mul_reduce(dxs, x, s, B, H * W, I)
        if ctx.needs_input_grad[1]:
            gw = core.flat_grad(wparam)
            acc = 0
            if gw is not None and gw.permute(0, 2, 3, 1).is_contiguous():
                dw_ohwi, acc = gw.permute(0, 2, 3, 1), 1
            else:
                dw_ohwi = torch.empty((O, KH, KW, I), device=dy.device, dtype=torch.float32)
            xt = core.tensor4_nhwc(x)
            sk = 0   # the library picks tile and split-K factor together
            xw, dudw, sw, dw_ = _prescaled_for_wgrad(x, dud, s, d, B, H)
            core.engine_call('ldetr_conv_transpose2d_bwd_weight_f32', 2.0 * B * H * W * O * KH * KW * I, lambda: core.check(core.lib().ldetr_conv_transpose2d_bwd_weight_f32(core.ptr(xw), ctypes.byref(xt), core.ptr(dudw), ctypes.byref(dudt),
                                                                        core.ptr(dw_ohwi), KH, KW, 2, 0, sk, core.ptr(sw), sw.stride(0) if sw is not None else 0,
                                                                        core.ptr(dw_), dw_.stride(0) if dw_ is not None else 0, acc, core.stream()), 'modconv_up_bwd_weight'), operands=(x, dud, dw_ohwi))
            dw = None if acc else _grad_to_oihw(dw_ohwi)
        return dx, dw, ds, ddemod, dbias, None, None, None


class _ToRGBFn(torch.autograd.Function):
    """y = conv1x1(x * s) + bias   (no demodulation, linear activation; 3 output channels)"""

    @staticmethod
    def forward(ctx, x, weight, styles, bias):
        core.require_gpu(x, weight, styles, bias)
        x = core.f32c(x); s = core.f32c(styles); b = core.f32c(bias)
        w = core.f32c(weight.reshape(weight.shape[0], -1))  # [3, C]
        B, H, W, C = x.shape
        O = w.shape[0]
        y = torch.empty((B, H, W, O), device=x.device, dtype=torch.float32)
        c4 = C // 4
        if O == 3 and C % 4 == 0 and C <= 512 and (c4 & (c4 - 1)) == 0 and s.stride(0) == C:
            # streaming kernel (csrc/misc_ops.hip): one read of x; on the engine the 3 output channels pad a 64-wide tile (144 us at 16 x 256^2 x 32)
            core.check(core.lib().ldetr_torgb_fwd_f32(core.ptr(x), core.ptr(w), core.ptr(s), core.ptr(b), core.ptr(y), B, H * W, C, core.stream()), 'torgb_fwd')
        else:
            xt = core.tensor4_nhwc(x)
            ep = core.epilogue(col_bias=b)
            core.engine_call('ldetr_conv2d_fwd_f32', 2.0 * B * H * W * O * C, lambda: core.check(core.lib().ldetr_conv2d_fwd_f32(core.ptr(x), ctypes.byref(xt), core.ptr(w), O, 1, 1, 1, 0, core.ptr(y), O, H, W,
                                                       core.ptr(s), s.stride(0), ctypes.byref(ep), core.stream()), 'torgb_fwd'), operands=(x, y, w))
        ctx.save_for_backward(x, w, s)
        ctx.wshape = weight.shape
        ctx.params = (weight, bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, s = ctx.saved_tensors
        B, H, W, C = x.shape
        O = w.shape[0]
        if O != 3:
            raise NotImplementedError('toRGB backward kernel is specialised for 3 colour channels')
        dy = core.f32c(dy)
        dx = torch.empty_like(x)
        dws = core.zeros((B, O, C), x.device)
        # weight / bias gradients go straight into the flat .grad views when there are any (atomics / += in the kernels): no temporaries, no
        # AccumulateGrad launches; ds and dw come out of one finish launch instead of two broadcast multiplies + two reductions
        wparam, bparam = ctx.params
        gw, gb = core.flat_grad(wparam), core.flat_grad(bparam)
        gw = gw.reshape(O, C) if (gw is not None and gw.is_contiguous()) else None
        dbias = gb if gb is not None else torch.zeros(O, device=x.device, dtype=torch.float32)
        dw = gw if gw is not None else torch.zeros((O, C), device=x.device, dtype=torch.float32)
        ds = torch.empty((B, C), device=x.device, dtype=torch.float32)
        core.check(core.lib().ldetr_torgb_bwd_f32(core.ptr(x), core.ptr(dy), core.ptr(w), core.ptr(s), core.ptr(dx), core.ptr(dws),
                                                  core.ptr(dbias), B, H * W, C, core.stream()), 'torgb_bwd')
        core.check(core.lib().ldetr_torgb_bwd_finish_f32(core.ptr(dws), core.ptr(s), core.ptr(w), B, C, core.ptr(dw), core.ptr(ds), core.stream()), 'torgb_bwd_finish')
        return dx, (None if gw is not None else dw.reshape(ctx.wshape)), ds, (None if gb is not None else dbias)


class _DemodFn(torch.autograd.Function):
    """dcoefs[b,o] = rsqrt(sum_{i,kh,kw} (w[o,i,kh,kw] * s[b,i])^2 + 1e-8)  (networks_stylegan2.py:57-61) as one launch
    (csrc/demod.hip: the sum over taps of w^2 once per output channel, then a dot product per sample) and two for the backward,
    the weight gradient going straight into the flat gradient buffer when there is one."""

    @staticmethod
    def forward(ctx, weight, styles):
        core.require_gpu(weight, styles)
        w = weight.detach()
        if w.dtype != torch.float32:
            w = w.float()
        s = core.f32c(styles.detach())
        O, I, KH, KW = w.shape
        B = s.shape[0]
        d = torch.empty((B, O), device=s.device, dtype=torch.float32)
        w2 = torch.empty((O, I), device=s.device, dtype=torch.float32)
        st = w.stride()
        core.check(core.lib().ldetr_demod_fwd_f32(core.ptr(w), st[0], st[1], st[2], st[3], core.ptr(s), core.ptr(d), core.ptr(w2),
                                                  B, O, I, KH, KW, 1e-8, core.stream()), 'demod_fwd')
        ctx.save_for_backward(w, s, d, w2)
        ctx.wparam = weight
        return d

    @staticmethod
    def backward(ctx, g):
        w, s, d, w2 = ctx.saved_tensors
        O, I, KH, KW = w.shape
        B = s.shape[0]
        g = core.f32c(g)
        need_w, need_s = ctx.needs_input_grad
        acc = core.flat_grad(ctx.wparam) if need_w else None
        accumulate = acc is not None and acc.stride() == w.stride()
        dw = acc if accumulate else (torch.empty_strided(w.shape, w.stride(), device=w.device, dtype=torch.float32) if need_w else None)
        ds = torch.empty_like(s) if need_s else None
        st = w.stride()
        core.check(core.lib().ldetr_demod_bwd_f32(core.ptr(w), st[0], st[1], st[2], st[3], core.ptr(s), core.ptr(d), core.ptr(w2), core.ptr(g),
                                                  core.ptr(dw), 1 if accumulate else 0, core.ptr(ds), B, O, I, KH, KW, core.stream()), 'demod_bwd')
        return (None if accumulate else dw), ds


def demod_coefs(weight, styles):
    """Demodulation coefficients [B, O] of a modulated convolution."""
    if styles.shape[0] > 64:       # the fused kernels keep one value per sample in LDS; larger micro-batches take the GEMM form
        from .linear import linear
        return (linear(styles.square(), weight.square().sum(dim=[2, 3])) + 1e-8).rsqrt()
    return _DemodFn.apply(weight, styles)


def modconv3x3(x, weight, styles, bias, act_alpha=0.2, act_gain=math.sqrt(2)):
    d = demod_coefs(weight, styles)
    return _ModConvFn.apply(x, weight, styles, d, bias, weight.shape[2] // 2, act_alpha, act_gain)


def modconv3x3_up2(x, weight, styles, bias, f, act_alpha=0.2, act_gain=math.sqrt(2)):
    d = demod_coefs(weight, styles)
    return _ModConvUpFn.apply(x, weight, styles, d, bias, f, act_alpha, act_gain)


def torgb(x, weight, styles, bias):
    return _ToRGBFn.apply(x, weight, styles, bias)
