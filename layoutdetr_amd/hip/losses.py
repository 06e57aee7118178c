"""The tail of a loss phase on the HIP kernels (csrc/layout_loss.hip): `combine` = the weighted sum of a phase's loss terms, its mean and the
backward of all of that as one launch per direction (training/loss.py:84-116, 146-218 run ~70 scalar-sized ATen launches per phase for it), and
`masked_mse` = F.mse_loss(a[valid], b[valid]) of the static-shape heads without the gather (networks_detr.py:314, loss.py:240, 245)."""
import ctypes

import torch

from . import core

IDENT, SOFTPLUS, SOFTPLUS_NEG, RATIO = 0, 1, 2, 3


class Term(object):
    """One loss term: `x` a scalar, a per-sample vector [B] or the (sum, count) pair of a cross entropy (fn = RATIO); value = weight * f(x).
    sum_reduce: the vector holds per-sample CONTRIBUTIONS to a scalar (their sum is the term); otherwise the term is per-sample and the phase
    averages it over the batch (`(sum of terms).mean()` broadcasts scalars: training/loss.py:213, 253).
    A 2-D x [R, B] with lists of R names / weights / sum_reduce flags is R terms that share one autograd input (the four layout terms of
    metrics.metric_layoutnet._LayoutLossesFn: selecting its rows as views would cost a fill + copy + add per row in the backward)."""

    def __init__(self, name, x, weight=1.0, fn=IDENT, sum_reduce=False):
        self.name, self.x, self.weight, self.fn, self.sum_reduce = name, x, weight, fn, sum_reduce

    def rows(self):
        if isinstance(self.name, (list, tuple)):
            R = len(self.name)
            return [(self.name[r], float(self.weight[r]), self.fn, bool(self.sum_reduce[r])) for r in range(R)]
        return [(self.name, float(self.weight), self.fn, bool(self.sum_reduce))]


def _tables(ptrs, ns, ws, fns, reds):
    K = len(ptrs)
    return ((ctypes.c_void_p * K)(*ptrs), (ctypes.c_float * K)(*ws), (ctypes.c_int * K)(*ns), (ctypes.c_int * K)(*fns), (ctypes.c_int * K)(*reds))


class _CombineFn(torch.autograd.Function):
    """forward(spec, *xs): spec[j] = [(weight, fn, sum_reduce)] per row of xs[j] -> (total, vals [K, ld], sums [K]); K = all rows."""

    @staticmethod
    def forward(ctx, spec, *xs):
        core.require_gpu(*xs)
        flat = [core.f32c(x.detach().reshape(len(rows), -1)) for x, rows in zip(xs, spec)]
        ptrs, ns, ws, fns, reds = [], [], [], [], []
        for x, rows in zip(flat, spec):
            n = x.shape[1]
            for r, (w, fn, red) in enumerate(rows):
                ptrs.append(x.data_ptr() + 4 * r * n); ns.append(n); ws.append(w); fns.append(fn); reds.append(1 if red else 0)
        K, ld = len(ptrs), max(2, max(ns))
        if K > 16:
            raise RuntimeError('loss combine: at most 16 terms per launch')
        dev = flat[0].device
        vals = torch.empty((K, ld), device=dev, dtype=torch.float32)
        sums = torch.empty(K, device=dev, dtype=torch.float32)
        total = torch.empty((), device=dev, dtype=torch.float32)
        t = _tables(ptrs, ns, ws, fns, reds)
        core.check(core.lib().ldetr_loss_combine_fwd_f32(t[0], t[1], t[2], t[3], t[4], K, ld, core.ptr(vals), core.ptr(sums), core.ptr(total), core.stream()), 'loss_combine_fwd')
        ctx.save_for_backward(*flat)
        ctx.cfg = (ptrs, ns, ws, fns, reds, ld, [x.shape for x in xs])
        ctx.mark_non_differentiable(vals, sums)
        return total, vals, sums

    @staticmethod
    def backward(ctx, g, _gv, _gs):
        flat = ctx.saved_tensors
        ptrs, ns, ws, fns, reds, ld, shapes = ctx.cfg
        K = len(ptrs)
        grads = torch.empty((K, ld), device=g.device, dtype=torch.float32)
        t = _tables(ptrs, ns, ws, fns, reds)
        g = core.f32c(g.reshape(1))
        core.check(core.lib().ldetr_loss_combine_bwd_f32(t[0], t[1], t[2], t[3], t[4], K, ld, core.ptr(g), core.ptr(grads), core.stream()), 'loss_combine_bwd')
        out, k = [], 0
        for x, shape in zip(flat, shapes):
            R, n = x.shape
            out.append(grads[k:k + R, :n].reshape(shape))
            k += R
        return (None,) + tuple(out)


def combine(terms, gain=1.0):
    """terms: list of Term -> (total * gain as a scalar with autograd, {name: weighted value(s) as the reference reports them}).
    total = mean over the batch of the sum of the terms (scalars broadcast)."""
    if gain == 0:
        raise ValueError('combine: gain must be non-zero (the reported values are recovered from the gain-weighted ones)')
    if not all(t.x.is_cuda for t in terms):
        raise RuntimeError('combine: every term must live in GPU memory (no CPU fallback); got ' + ', '.join(str(t.x.device) for t in terms))
    if sum(len(t.rows()) for t in terms) > 16:
        raise ValueError(f'combine: at most 16 rows per launch (csrc/layout_loss.hip), got {sum(len(t.rows()) for t in terms)}: split the phase tail')
    spec = [[(w * gain, fn, red) for _, w, fn, red in t.rows()] for t in terms]
    total, vals, sums = _CombineFn.apply(spec, *[t.x for t in terms])
    inv = 1.0 / gain if gain != 1.0 else 1.0
    rep, k = {}, 0
    for t in terms:
        rows = t.rows()
        n = t.x.numel() // len(rows)
        for name, _, fn, red in rows:
            v = sums[k] if (red or fn == RATIO or n == 1) else vals[k, :n]
            rep[name] = v if inv == 1.0 else v * inv
            k += 1
    return total, rep


class _MaskedMseFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, valid_u8, bdiv):
        core.require_gpu(a, b, valid_u8)
        D = a.shape[-1]
        a2, b2 = core.f32c(a.reshape(-1, D)), core.f32c(b.detach().reshape(-1, D))
        rows = a2.shape[0]
        assert valid_u8.numel() == rows and b2.shape[0] * bdiv == rows
        out = torch.empty(2, device=a.device, dtype=torch.float32)
        core.check(core.lib().ldetr_masked_mse_fwd_f32(core.ptr(a2), core.ptr(b2), core.ptr(valid_u8), rows, D, bdiv, core.ptr(out), core.stream()), 'masked_mse_fwd')
        ctx.save_for_backward(a2, b2, valid_u8, out)
        ctx.cfg = (rows, D, bdiv, a.shape)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        a2, b2, valid_u8, out = ctx.saved_tensors
        rows, D, bdiv, shape = ctx.cfg
        da = torch.empty_like(a2)
        g = core.f32c(g.reshape(1))
        core.check(core.lib().ldetr_masked_mse_bwd_f32(core.ptr(a2), core.ptr(b2), core.ptr(valid_u8), rows, D, bdiv, core.ptr(out), core.ptr(g), core.ptr(da), core.stream()),
                   'masked_mse_bwd')
        return da.reshape(shape), None, None, None


def masked_mse(a, b, valid_u8, bdiv=1):
    """a [..., D] (gradient), b [rows / bdiv, D] (data: no gradient), valid_u8 [rows] uint8 -> scalar F.mse_loss(a[valid], b[valid])."""
    if b.requires_grad:
        raise NotImplementedError('masked_mse: the reference tensor is data (no gradient is produced for it)')
    return _MaskedMseFn.apply(a, b, valid_u8.reshape(-1), bdiv)
