"""Fused bias + activation, signature-compatible with the reference `torch_utils/ops/bias_act.py`.

`bias_act(x, b, dim, act, alpha, gain, clamp, impl)` keeps the reference semantics
(bias_act.py:53-88) and its first/second-order autograd structure (bias_act.py:127-206), but every
evaluation is the gfx950 kernel behind `ldetr_bias_act_f32` (C ABI, include/ldetr_hip.h).  There is
no `_ref` fallback on this path: non-GPU tensors raise.
"""
import numpy as np
import torch

from ...hip import core


class _Spec(dict):
    __getattr__ = dict.__getitem__


# name -> (default alpha, default gain, kernel index, which tensors backward needs, has 2nd derivative)
activation_funcs = {
    'linear':   _Spec(def_alpha=0,   def_gain=1,          cuda_idx=1, ref='',  has_2nd_grad=False),
    'relu':     _Spec(def_alpha=0,   def_gain=np.sqrt(2), cuda_idx=2, ref='y', has_2nd_grad=False),
    'lrelu':    _Spec(def_alpha=0.2, def_gain=np.sqrt(2), cuda_idx=3, ref='y', has_2nd_grad=False),
    'tanh':     _Spec(def_alpha=0,   def_gain=1,          cuda_idx=4, ref='y', has_2nd_grad=True),
    'sigmoid':  _Spec(def_alpha=0,   def_gain=1,          cuda_idx=5, ref='y', has_2nd_grad=True),
    'elu':      _Spec(def_alpha=0,   def_gain=1,          cuda_idx=6, ref='y', has_2nd_grad=True),
    'selu':     _Spec(def_alpha=0,   def_gain=1,          cuda_idx=7, ref='y', has_2nd_grad=True),
    'softplus': _Spec(def_alpha=0,   def_gain=1,          cuda_idx=8, ref='y', has_2nd_grad=True),
    'swish':    _Spec(def_alpha=0,   def_gain=np.sqrt(2), cuda_idx=9, ref='x', has_2nd_grad=True),
}


def _dense_layout(x):
    """Return x (possibly copied) in a dense layout plus the element stride of `dim`-indexed bias steps."""
    if x.ndim > 2 and x.stride(1) == 1 and x.is_contiguous(memory_format=torch.channels_last):
        return x, torch.channels_last
    return x.contiguous(), torch.contiguous_format


def _launch(x, b, xref, yref, dy, grad, dim, spec, alpha, gain, clamp):
    core.require_gpu(x, b, xref, yref, dy)
    y = torch.empty_like(x)  # preserves the dense layout of x
    step_b = x.stride(dim) if b is not None else 1
    core.check(core.lib().ldetr_bias_act_f32(
        core.ptr(x), core.ptr(b), core.ptr(xref), core.ptr(yref), core.ptr(dy), core.ptr(y), x.numel(),
        b.numel() if b is not None else 0, step_b, grad, spec.cuda_idx, alpha, gain, clamp, core.stream()), 'bias_act')
    return y


_cache = dict()


def _make(dim, act, alpha, gain, clamp):
    key = (dim, act, alpha, gain, clamp)
    if key in _cache:
        return _cache[key]
    spec = activation_funcs[act]

    class BiasAct(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, b):
            if x.dtype != torch.float32:
                raise RuntimeError('bias_act: only float32 is implemented on the gfx950 path')
            x, ctx.memory_format = _dense_layout(x)
            b = b.contiguous() if b is not None else None
            if b is not None and (b.ndim != 1 or b.shape[0] != x.shape[dim]):
                raise RuntimeError('bias_act: b has wrong number of elements')
            y = x
            if act != 'linear' or gain != 1 or clamp >= 0 or b is not None:
                y = _launch(x, b, None, None, None, 0, dim, spec, alpha, gain, clamp)
            need_x = 'x' in spec.ref or spec.has_2nd_grad
            ctx.save_for_backward(x if need_x else None, b if need_x else None, y if 'y' in spec.ref else None)
            return y

        @staticmethod
        def backward(ctx, dy):
            dy = dy.contiguous(memory_format=ctx.memory_format)
            x, b, y = ctx.saved_tensors
            dx = db = None
            if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
                dx = dy
                if act != 'linear' or gain != 1 or clamp >= 0:
                    dx = BiasActGrad.apply(dy, x, b, y)
            if ctx.needs_input_grad[1]:
                db = dx.sum([i for i in range(dx.ndim) if i != dim])
            return dx, db

    class BiasActGrad(torch.autograd.Function):
        @staticmethod
        def forward(ctx, dy, x, b, y):
            ctx.memory_format = torch.channels_last if dy.ndim > 2 and dy.stride(1) == 1 else torch.contiguous_format
            dx = _launch(dy, b, x, y, None, 1, dim, spec, alpha, gain, clamp)
            ctx.save_for_backward(dy if spec.has_2nd_grad else None, x, b, y)
            return dx

        @staticmethod
        def backward(ctx, d_dx):
            d_dx = d_dx.contiguous(memory_format=ctx.memory_format)
            dy, x, b, y = ctx.saved_tensors
            d_dy = d_x = d_b = None
            if ctx.needs_input_grad[0]:
                d_dy = BiasActGrad.apply(d_dx, x, b, y)
            if spec.has_2nd_grad and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]):
                d_x = _launch(d_dx, b, x, y, dy, 2, dim, spec, alpha, gain, clamp)
            if spec.has_2nd_grad and ctx.needs_input_grad[2]:
                d_b = d_x.sum([i for i in range(d_x.ndim) if i != dim])
            return d_dy, d_x, d_b, None

    _cache[key] = BiasAct
    return BiasAct


def bias_act(x, b=None, dim=1, act='linear', alpha=None, gain=None, clamp=None, impl='cuda'):
    assert isinstance(x, torch.Tensor)
    assert impl in ['ref', 'cuda']
    assert clamp is None or clamp >= 0
    if impl == 'ref':
        raise RuntimeError("bias_act: impl='ref' does not exist in layoutdetr_amd (the CPU restatement lives in oracle/)")
    spec = activation_funcs[act]
    alpha = float(alpha if alpha is not None else spec.def_alpha)
    gain = float(gain if gain is not None else spec.def_gain)
    clamp = float(clamp if clamp is not None else -1)
    if b is not None:
        assert isinstance(b, torch.Tensor) and b.ndim == 1
        assert 0 <= dim < x.ndim
        assert b.shape[0] == x.shape[dim]
    return _make(dim, act, alpha, gain, clamp).apply(x, b)
