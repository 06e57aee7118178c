"""upfirdn2d family, signature-compatible with the reference `torch_utils/ops/upfirdn2d.py`
(setup_filter :67-115, upfirdn2d :119-166, filter2d/upsample2d/downsample2d :278-390), evaluated by
`ldetr_upfirdn2d_f32`.  Gradients of arbitrary order come from re-entrant autograd Functions exactly
as in the reference (:252-270): the backward of an upfirdn2d is another upfirdn2d.
"""
import numpy as np
import torch

from ...hip import core


def _parse_scaling(scaling):
    if isinstance(scaling, int):
        scaling = [scaling, scaling]
    assert isinstance(scaling, (list, tuple)) and all(isinstance(v, int) for v in scaling)
    sx, sy = scaling
    assert sx >= 1 and sy >= 1
    return sx, sy


def _parse_padding(padding):
    if isinstance(padding, int):
        padding = [padding, padding]
    assert isinstance(padding, (list, tuple)) and all(isinstance(v, int) for v in padding)
    if len(padding) == 2:
        px, py = padding
        padding = [px, px, py, py]
    px0, px1, py0, py1 = padding
    return px0, px1, py0, py1


def _get_filter_size(f):
    if f is None:
        return 1, 1
    assert isinstance(f, torch.Tensor) and f.ndim in [1, 2]
    return int(f.shape[-1]), int(f.shape[0])


def setup_filter(f, device=torch.device('cpu'), normalize=True, flip_filter=False, gain=1, separable=None):
    """Build the FIR tap tensor: 1-D taps with < 8 entries are expanded to their outer product."""
    if f is None:
        f = 1
    f = torch.as_tensor(f, dtype=torch.float32)
    assert f.ndim in [0, 1, 2] and f.numel() > 0
    if f.ndim == 0:
        f = f[np.newaxis]
    if separable is None:
        separable = (f.ndim == 1 and f.numel() >= 8)
    if f.ndim == 1 and not separable:
        f = f.ger(f)
    assert f.ndim == (1 if separable else 2)
    if normalize:
        f = f / f.sum()
    if flip_filter:
        f = f.flip(list(range(f.ndim)))
    f = f * (gain ** (f.ndim / 2))
    return f.to(device=device)


def _kernel_call(x, f2d, upx, upy, downx, downy, px0, px1, py0, py1, flip, gain, act_bias=None, act=None):
    """One launch.  x: [N,C,H,W] fp32 (any strides), f2d: [fh,fw] fp32 on the same device."""
    core.require_gpu(x, f2d)
    if x.dtype != torch.float32:
        raise RuntimeError('upfirdn2d: only float32 is implemented on the gfx950 path')
    if f2d.dtype != torch.float32:
        raise RuntimeError('upfirdn2d: f must be float32')
    N, C, H, W = x.shape
    fh, fw = f2d.shape
    outW = (W * upx + px0 + px1 - fw + downx) // downx
    outH = (H * upy + py0 + py1 - fh + downy) // downy
    if outW < 1 or outH < 1:
        raise RuntimeError('upfirdn2d: output must be at least 1x1')
    cl = x.ndim == 4 and x.stride(1) == 1 and C > 1
    y = torch.empty((N, C, outH, outW), device=x.device, dtype=torch.float32,
                    memory_format=torch.channels_last if cl else torch.contiguous_format)
    import ctypes
    xs = (ctypes.c_int64 * 4)(*x.stride())
    ys = (ctypes.c_int64 * 4)(*y.stride())
    has_act, a_alpha, a_gain = (0, 0.0, 1.0) if act is None else (1, float(act[0]), float(act[1]))
    core.check(core.lib().ldetr_upfirdn2d_f32(
        core.ptr(x), core.ptr(f2d), core.ptr(y), N, C, H, W, xs, fh, fw, f2d.stride(0), f2d.stride(1), upx, upy, downx,
        downy, px0, px1, py0, py1, 1 if flip else 0, float(gain), outH, outW, ys, core.ptr(act_bias), has_act, a_alpha,
        a_gain, core.stream()), 'upfirdn2d')
    return y


_cache = dict()


def _make(up, down, padding, flip_filter, gain):
    upx, upy = _parse_scaling(up)
    downx, downy = _parse_scaling(down)
    px0, px1, py0, py1 = _parse_padding(padding)
    key = (upx, upy, downx, downy, px0, px1, py0, py1, flip_filter, gain)
    if key in _cache:
        return _cache[key]

    class Upfirdn2d(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, f):
            assert isinstance(x, torch.Tensor) and x.ndim == 4
            if f is None:
                f = torch.ones([1, 1], dtype=torch.float32, device=x.device)
            if f.ndim == 1 and f.shape[0] == 1:
                f = f.square().unsqueeze(0)
            assert f.ndim in [1, 2]
            if f.ndim == 2:
                y = _kernel_call(x, f, upx, upy, downx, downy, px0, px1, py0, py1, flip_filter, gain)
            else:
                y = _kernel_call(x, f.unsqueeze(0), upx, 1, downx, 1, px0, px1, 0, 0, flip_filter, 1.0)
                y = _kernel_call(y, f.unsqueeze(1), 1, upy, 1, downy, 0, 0, py0, py1, flip_filter, gain)
            ctx.save_for_backward(f)
            ctx.x_shape = x.shape
            return y

        @staticmethod
        def backward(ctx, dy):
            f, = ctx.saved_tensors
            _, _, ih, iw = ctx.x_shape
            _, _, oh, ow = dy.shape
            fw, fh = _get_filter_size(f)
            p = [fw - px0 - 1, iw * upx - ow * downx + px0 - upx + 1, fh - py0 - 1, ih * upy - oh * downy + py0 - upy + 1]
            dx = None
            if ctx.needs_input_grad[0]:
                dx = _make(up=[downx, downy], down=[upx, upy], padding=p, flip_filter=(not flip_filter), gain=gain).apply(dy, f)
            assert not ctx.needs_input_grad[1]
            return dx, None

    _cache[key] = Upfirdn2d
    return Upfirdn2d


def upfirdn2d(x, f, up=1, down=1, padding=0, flip_filter=False, gain=1, impl='cuda'):
    assert isinstance(x, torch.Tensor)
    assert impl in ['ref', 'cuda']
    if impl == 'ref':
        raise RuntimeError("upfirdn2d: impl='ref' does not exist in layoutdetr_amd (the CPU restatement lives in oracle/)")
    return _make(up=up, down=down, padding=padding, flip_filter=flip_filter, gain=gain).apply(x, f)


def filter2d(x, f, padding=0, flip_filter=False, gain=1, impl='cuda'):
    px0, px1, py0, py1 = _parse_padding(padding)
    fw, fh = _get_filter_size(f)
    p = [px0 + fw // 2, px1 + (fw - 1) // 2, py0 + fh // 2, py1 + (fh - 1) // 2]
    return upfirdn2d(x, f, padding=p, flip_filter=flip_filter, gain=gain, impl=impl)


def upsample2d(x, f, up=2, padding=0, flip_filter=False, gain=1, impl='cuda'):
    upx, upy = _parse_scaling(up)
    px0, px1, py0, py1 = _parse_padding(padding)
    fw, fh = _get_filter_size(f)
    p = [px0 + (fw + upx - 1) // 2, px1 + (fw - upx) // 2, py0 + (fh + upy - 1) // 2, py1 + (fh - upy) // 2]
    return upfirdn2d(x, f, up=up, padding=p, flip_filter=flip_filter, gain=gain * upx * upy, impl=impl)


def downsample2d(x, f, down=2, padding=0, flip_filter=False, gain=1, impl='cuda'):
    downx, downy = _parse_scaling(down)
    px0, px1, py0, py1 = _parse_padding(padding)
    fw, fh = _get_filter_size(f)
    p = [px0 + (fw - downx + 1) // 2, px1 + (fw - downx) // 2, py0 + (fh - downy + 1) // 2, py1 + (fh - downy) // 2]
    return upfirdn2d(x, f, down=down, padding=p, flip_filter=flip_filter, gain=gain, impl=impl)
