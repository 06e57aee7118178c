"""nn.Linear-shaped contractions on the f32 MFMA GEMM engine (forward, dX, dW, fused bias/act/dropout).

Replaces ATen addmm/matmul behind every nn.Linear on the hot path: FFN 256<->2048
(training/detr_transformer.py:187-189,212), MLP heads (training/networks_detr.py:50-62) and
FullyConnectedLayer (training/networks_stylegan2.py:117-123).
"""
import ctypes
import os

import torch

from . import core
from .core import ACT_LRELU, ACT_NONE, ACT_RELU


def act_backward(dy2, y2, act, act_alpha, act_gain, want_dbias, bias=None, demod=None, want_ddemod=False, B=1, dbias_out=None):
    """dv = dy * gain * dact(y); optional fused reductions.  2-D [rows, C] tensors.
    dbias_out: existing [C] buffer to accumulate the bias gradient into (returned dbias is then None)."""
    R, C = dy2.shape
    if C % 4 != 0:
        if act == ACT_NONE:
            dv = dy2 * act_gain if act_gain != 1.0 else dy2
        else:
            neg = 0.0 if act == ACT_RELU else act_alpha
            dv = dy2 * torch.where(y2 > 0, act_gain, act_gain * neg)
        if want_dbias and dbias_out is not None:
            dbias_out += dv.sum(0)
            return dv, None, None
        return dv, (dv.sum(0) if want_dbias else None), None
    dv = torch.empty_like(dy2)
    if want_dbias and dbias_out is not None:
        dbias = dbias_out
    else:
        dbias = torch.zeros(C, device=dy2.device, dtype=torch.float32) if want_dbias else None      # (returned to autograd as a parameter gradient: never arena memory)
    ddemod = core.zeros((B, C), dy2.device) if want_ddemod else None
    core.check(core.lib().ldetr_act_bwd_reduce_f32(
        core.ptr(dy2), core.ptr(y2), core.ptr(dv), core.ptr(bias), core.ptr(demod), core.ptr(dbias), core.ptr(ddemod),
        B, R // B, C, act, act_alpha, act_gain, core.stream()), 'act_bwd_reduce')
    return dv, (None if dbias_out is not None else dbias), ddemod


class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, act, act_alpha, act_gain, p_drop, wscale, rows, add_input, passthru):
        core.require_gpu(x, weight, bias)
        ctx.set_materialize_grads(False)
        r0, r1 = rows if rows is not None else (0, weight.shape[0])   # row range of a packed weight (MHA in_proj)
        N, K = r1 - r0, weight.shape[1]
        x2 = x.reshape(-1, K)
        if not (x2.dtype == torch.float32 and x2.stride(1) == 1 and x2.stride(0) >= K and x2.stride(0) % 4 == 0 and x2.data_ptr() % 16 == 0):
            x2 = core.f32c(x2)        # (rows with a pitch, e.g. ws[:, i] of the StyleGAN2 mapping output or the layout token x[0], go to the GEMM as they are: lda)
        if add_input is not None:          # y = f((x + add_input) W^T): the position embedding of q/k (no gradient for it)
            if add_input.requires_grad:
                raise RuntimeError('linear: add_input is a constant (the sine position embedding); a learned embedding must be added by the caller')
            x2 = x2 + add_input.reshape(-1, K)
        w = core.f32c(weight.detach()[r0:r1])
        b = core.f32c(bias.detach()[r0:r1]) if bias is not None else None
        M = x2.shape[0]
        seed = core.next_seed() if p_drop > 0 else 0
        ep = core.epilogue(alpha=wscale, col_bias=b, act=act, act_alpha=act_alpha, act_gain=act_gain, p_drop=p_drop,
                           seed=seed)
        y = core.gemm(x2, w, 0, 0, M, N, K, ep=ep)
        ctx.save_for_backward(x2, w, y if act != ACT_NONE else None)
        ctx.cfg = (act, act_alpha, act_gain, p_drop, wscale, bias is not None, x.shape)
        ctx.params = (weight, bias, r0, r1)
        if p_drop > 0 and act != ACT_RELU:
            raise RuntimeError('linear: fused dropout is only defined after relu (FFN hidden layer)')
        if passthru:
            # second output = x itself (autograd aliases it).  A residual connection that reads x through THIS alias makes the
            # node the only consumer of x: its backward receives the residual-path gradient and adds it in the dX GEMM's
            # epilogue, instead of autograd launching an add kernel to sum two gradients of x.
            return y.reshape(*x.shape[:-1], N), x
        return y.reshape(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy, dx_pass=None):
        x2, w, y = ctx.saved_tensors
        act, act_alpha, act_gain, p_drop, wscale, has_bias, xshape = ctx.cfg
        N, K = w.shape
        M = x2.shape[0]
        if dy is None:                      # only the pass-through output was used
            return (dx_pass,) + (None,) * 10
        dy2 = core.f32c(dy.reshape(-1, N))
        need_x, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], has_bias and ctx.needs_input_grad[2]
        wparam, bparam, r0, r1 = ctx.params
        gw = core.flat_grad(wparam) if need_w else None
        gb = core.flat_grad(bparam) if need_b else None
        if gw is not None:
            gw = gw[r0:r1]
        if gb is not None:
            gb = gb[r0:r1] if (N % 4 == 0 and r0 % 4 == 0) else None
        pairable = need_x and need_w and gw is not None and gw.is_contiguous()
        dx = dw = db = None
        if act != ACT_NONE:
            gain = act_gain / (1.0 - p_drop) if p_drop > 0 else act_gain
            dpre, db, _ = act_backward(dy2, y, act, act_alpha, gain, need_b, dbias_out=gb)
        else:
            dpre = dy2
            db = None
            fold_db = need_b and gb is not None and need_w and gw is not None and gw.is_contiguous() and dpre.is_contiguous()
            if need_b and not fold_db:
                if gb is not None:
                    core.check(core.lib().ldetr_colsum_f32(core.ptr(dpre), core.ptr(gb), 1, M, N, core.stream()), 'colsum')
                else:
                    db = core.colsum(dpre).reshape(-1)
        if need_x and need_w and pairable:
            # dX = dY W and dW += dY^T X through one C-ABI call: one kernel launch when both are small-tile problems
            res = core.f32c(dx_pass.reshape(-1, K)) if dx_pass is not None else None
            dx2 = torch.empty((M, K), device=dpre.device, dtype=torch.float32)
            rsum = gb if (act == ACT_NONE and fold_db) else None
            core.gemm_pair(dict(A=dpre, B=w, ta=0, tb=1, M=M, N=K, K=N, out=dx2, ep=core.epilogue(alpha=wscale, residual=res)),
                           dict(A=dpre, B=x2, ta=1, tb=1, M=N, N=K, K=M, out=gw, ep=core.epilogue(alpha=wscale, accumulate=True, a_rowsum=rsum)))
            dx = dx2.reshape(xshape)
            need_x = need_w = False
        if need_x:
            res = core.f32c(dx_pass.reshape(-1, K)) if dx_pass is not None else None
            dx = core.gemm(dpre, w, 0, 1, M, K, N, ep=core.epilogue(alpha=wscale, residual=res)).reshape(xshape)
        if need_w:
            if gw is not None and gw.is_contiguous():
                # dW = dY^T X accumulated into the flat .grad; the bias gradient (column sums of dY) rides along (a_rowsum)
                rsum = gb if (act == ACT_NONE and fold_db) else None
                core.gemm(dpre, x2, 1, 1, N, K, M, out=gw, ep=core.epilogue(alpha=wscale, accumulate=True, a_rowsum=rsum))
            else:
                dw = core.gemm(dpre, x2, 1, 1, N, K, M, ep=core.epilogue(alpha=wscale))
        full = wparam.shape[0]
        if dw is not None and (r1 - r0) != full:      # fallback path for a packed weight without a flat .grad
            dwf = torch.zeros((full, K), device=dw.device, dtype=torch.float32); dwf[r0:r1] = dw; dw = dwf
        if db is not None and (r1 - r0) != full:
            dbf = torch.zeros(full, device=db.device, dtype=torch.float32); dbf[r0:r1] = db; db = dbf
        return dx, dw, db, None, None, None, None, None, None, None, None


def linear(x, weight, bias=None, act=ACT_NONE, act_alpha=0.0, act_gain=1.0, p_drop=0.0, wscale=1.0, rows=None, add_input=None,
           passthru=False):
    """y = dropout(act(((x [+ add_input]) @ (wscale*weight[rows]).T) + bias[rows]) * act_gain);  rows=(r0, r1) selects a row block
    of a packed projection weight (nn.MultiheadAttention.in_proj_weight) without creating an autograd slice node.
    passthru=True returns (y, x_alias): route the residual branch through x_alias (see _LinearFn.forward)."""
    return _LinearFn.apply(x, weight, bias, act, act_alpha, act_gain, p_drop, wscale, rows, add_input, passthru)
