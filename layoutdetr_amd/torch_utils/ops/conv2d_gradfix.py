"""`conv2d_gradfix` surface of the reference (torch_utils/ops/conv2d_gradfix.py:22-58): `conv2d`, `conv_transpose2d`,
`no_weight_gradients`, `enabled`, `weight_gradients_disabled` — evaluated by the f32-MFMA implicit-GEMM engine
(`ldetr_conv2d_*_f32`, `ldetr_conv_transpose2d_*_f32`) instead of cuDNN.  Arguments and layouts are torch's: NCHW-shaped
tensors (channels_last memory is consumed without a copy), conv weight [O, I, kh, kw], transposed-conv weight [I, O, kh, kw].
Restrictions of the gfx950 path (everything the hot path uses): groups == 1, dilation == 1, output_padding == 0, symmetric
padding, fp32, input channels a multiple of 4 for the transposed form.  There is no CPU fallback.
"""
import contextlib

import torch

from ...hip import conv as hconv
from ...hip import core

enabled = True                      # kept for API compatibility: the engine is always the implementation here
weight_gradients_disabled = False   # mirror of core.WEIGHT_GRADIENTS_DISABLED[0]


@contextlib.contextmanager
def no_weight_gradients(disable=True):
    """Inside the block the conv Functions skip their weight gradients (used by the R1 / path-length regularisers,
    training/loss.py:132,210)."""
    global weight_gradients_disabled
    old = core.WEIGHT_GRADIENTS_DISABLED[0]
    if disable:
        core.WEIGHT_GRADIENTS_DISABLED[0] = True
        weight_gradients_disabled = True
    try:
        yield
    finally:
        core.WEIGHT_GRADIENTS_DISABLED[0] = old
        weight_gradients_disabled = old


def _pair(v, name):
    if isinstance(v, int):
        return v, v
    v = tuple(int(t) for t in v)
    if len(v) == 1:
        return v[0], v[0]
    if len(v) != 2:
        raise ValueError(f'{name} must be an int or a pair')
    return v


def _check(dilation, groups, sy, sx, py, px):
    if _pair(dilation, 'dilation') != (1, 1) or groups != 1:
        raise NotImplementedError('conv2d_gradfix: only dilation 1 / groups 1 are implemented on the gfx950 path')
    if sy != sx or py != px:
        raise NotImplementedError('conv2d_gradfix: stride and padding must be equal along both axes')


def _pad4(t, dim):
    """zero-pad `dim` to a multiple of 4 (the engine's vector width); autograd slices the gradient back."""
    extra = (-t.shape[dim]) % 4
    if extra == 0:
        return t
    shape = list(t.shape); shape[dim] = extra
    return torch.cat([t, t.new_zeros(shape)], dim)


def conv2d(input, weight, bias=None, stride=1, padding=0, dilation=1, groups=1):
    core.require_gpu(input, weight, bias)
    (sy, sx), (py, px) = _pair(stride, 'stride'), _pair(padding, 'padding')
    _check(dilation, groups, sy, sx, py, px)
    O = weight.shape[0]
    # channel counts that are not multiples of 4 (RGB heads) run zero-padded: generic callers only, the hot path never needs it
    x, w = _pad4(input, 1), _pad4(_pad4(weight, 1), 0)
    b = _pad4(bias, 0) if bias is not None else None
    y = hconv.conv2d_nhwc(x.permute(0, 2, 3, 1), w, None, b, None, stride=sy, pad=py)
    return y.permute(0, 3, 1, 2)[:, :O]


def conv_transpose2d(input, weight, bias=None, stride=1, padding=0, output_padding=0, groups=1, dilation=1):
    core.require_gpu(input, weight, bias)
    (sy, sx), (py, px) = _pair(stride, 'stride'), _pair(padding, 'padding')
    _check(dilation, groups, sy, sx, py, px)
    if _pair(output_padding, 'output_padding') != (0, 0):
        raise NotImplementedError('conv2d_gradfix.conv_transpose2d: output_padding must be 0')
    O = weight.shape[1]
    x, w = _pad4(input, 1), _pad4(_pad4(weight, 0), 1)
    y = hconv.conv_transpose2d_nhwc(x.permute(0, 2, 3, 1), w.permute(1, 2, 3, 0), stride=sy, pad=py)[..., :O]
    if bias is not None:
        y = y + bias
    return y.permute(0, 3, 1, 2)
