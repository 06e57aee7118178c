"""Twice-differentiable path through the layout heads and decoder stacks, for the regulariser phases only.

R1 (training/loss.py:207-215) differentiates D's score by the real boxes and path length (loss.py:119-142) G's boxes by the latents, both
with `create_graph=True`, and then back-propagate a function of that gradient: every node between (boxes | latents) and (score | boxes) needs a
backward that autograd can differentiate again.  The fused Functions of the hot path (hip/stacks.py, hip/attention.py, hip/ffn.py,
hip/linear.py) call kernels in their backward and are first-order by construction (SURVEY §7 "second-order autograd": fall back to a
composite inside such a phase).  While `higher_order()` is active the modules on those two paths -- `Linear` / `MLP` heads and
`TransformerDecoder` (networks_detr.py, detr_transformer.py) -- route here instead:

* every contraction is an autograd node over this package's own first-order kernels whose backward is the SAME node type on transposed
  views -- `mm` (csrc/gemm_conv.hip through `core.gemm`) for projections / feed-forward / heads, `bmm4` (csrc/bmm_strided.hip) for the
  per-(sample, head) attention products -- so derivatives of any order stay on these kernels;
* softmax, LayerNorm statistics, ReLU, dropout, bias / residual adds are torch element-wise / row ops (differentiable to any order).

The ResNet trunk, `input_proj` and the image encoder are not between the differentiated input and output (they produce the decoder's
memory): they keep their fused first-order Functions and receive the regulariser's gradient through the memory like any other loss.
Cost is not a concern here: the phases run every 16th / 4th iteration (lazy regularisation, training_loop.py:186-197) on 9-10 tokens per sample.
"""
import ctypes
import math

import torch
import torch.nn.functional as F

from . import core

_DEPTH = [0]


class higher_order(object):
    """`with higher_order():` -- modules built inside route their decoder-side sub-graph through this file."""

    def __enter__(self):
        _DEPTH[0] += 1
        return self

    def __exit__(self, *exc):
        _DEPTH[0] -= 1
        return False


def active():
    return _DEPTH[0] > 0


def _rowmajor(x):
    """2-D view -> (row-major tensor to hand to the GEMM, its leading dimension, 1 if that tensor is x transposed)."""
    r, c = x.shape
    if c == 1 or x.stride(1) == 1:
        ld = x.stride(0) if r > 1 else c
        if ld >= c:
            return x, ld, 0
    if r == 1 or x.stride(0) == 1:
        ld = x.stride(1) if c > 1 else r
        if ld >= r:
            return x.t(), ld, 1
    return x.contiguous(), c, 0


class _MmFn(torch.autograd.Function):
    """C = A @ B on 2-D fp32 views (row- or column-major, no copy); backward = two more `_MmFn` nodes."""

    @staticmethod
    def forward(ctx, A, B):
        core.require_gpu(A, B)
        M, K = A.shape
        K2, N = B.shape
        assert K == K2, (A.shape, B.shape)
        a, lda, at = _rowmajor(A.detach())     # at: `a` is A^T stored [K, M]  -> ta = 1
        b, ldb, bt = _rowmajor(B.detach())     # bt: `b` is B^T stored [N, K]  -> tb = 0 ; plain B [K, N] -> tb = 1
        out = torch.empty((M, N), device=A.device, dtype=torch.float32)
        if M and N:
            if K:
                core.gemm(a, b, at, 0 if bt else 1, M, N, K, out=out, lda=lda, ldb=ldb)
            else:
                out.zero_()
        ctx.save_for_backward(A, B)
        return out

    @staticmethod
    def backward(ctx, g):
        A, B = ctx.saved_tensors
        dA = mm(g, B.t()) if ctx.needs_input_grad[0] else None
        dB = mm(A.t(), g) if ctx.needs_input_grad[1] else None
        return dA, dB


def mm(A, B):
    return _MmFn.apply(A.float(), B.float())


def _strides4(x):
    return (ctypes.c_int64 * 4)(*[int(s) for s in x.stride()])


class _Bmm4Fn(torch.autograd.Function):
    """C[b1, b2] = alpha * A[b1, b2] @ B[b1, b2] on 4-D strided views (csrc/bmm_strided.hip); backward = two more `_Bmm4Fn` nodes."""

    @staticmethod
    def forward(ctx, A, B, alpha):
        core.require_gpu(A, B)
        n1, n2, M, K = A.shape
        assert B.shape[:2] == A.shape[:2] and B.shape[2] == K, (A.shape, B.shape)
        N = B.shape[3]
        out = torch.empty((n1, n2, M, N), device=A.device, dtype=torch.float32)
        a, b = A.detach(), B.detach()
        core.check(core.lib().ldetr_bmm_strided_f32(core.ptr(a), _strides4(a), core.ptr(b), _strides4(b), core.ptr(out), _strides4(out),
                                                    n1, n2, M, N, K, float(alpha), core.stream()), 'bmm_strided')
        ctx.save_for_backward(A, B)
        ctx.alpha = float(alpha)
        return out

    @staticmethod
    def backward(ctx, g):
        A, B = ctx.saved_tensors
        dA = bmm4(g, B.transpose(2, 3), ctx.alpha) if ctx.needs_input_grad[0] else None
        dB = bmm4(A.transpose(2, 3), g, ctx.alpha) if ctx.needs_input_grad[1] else None
        return dA, dB, None


def bmm4(A, B, alpha=1.0):
    return _Bmm4Fn.apply(A, B, alpha)


def linear(x, weight, bias=None, relu=False):
    """F.linear (+ ReLU) over `mm`."""
    K = weight.shape[1]
    y = mm(x.reshape(-1, K), weight.t())
    if bias is not None:
        y = y + bias
    if relu:
        y = torch.relu(y)
    return y.reshape(*x.shape[:-1], weight.shape[0])


def layer_norm(x, norm):
    """nn.LayerNorm over the last dimension, written out (biased variance, eps inside the square root)."""
    mu = x.mean(-1, keepdim=True)
    xc = x - mu
    var = (xc * xc).mean(-1, keepdim=True)
    return xc * torch.rsqrt(var + norm.eps) * norm.weight + norm.bias


def mha(q_in, k_in, v_in, m, B, Lq, Lk, kpm, training):
    """nn.MultiheadAttention as the reference calls it (detr_transformer.py:272-280; packed in_proj_weight, 1/sqrt(dh) on q, bool key-padding
    mask -> -inf, dropout on the probabilities, out_proj) on batch-first rows: q_in [B*Lq, d], k_in / v_in [B*Lk, d] -> [B*Lq, d]."""
    d, H = q_in.shape[1], m.num_heads
    dh = d // H
    W, b = m.in_proj_weight, m.in_proj_bias
    q = linear(q_in, W[:d], b[:d]).view(B, Lq, H, dh).permute(0, 2, 1, 3)              # [B, H, Lq, dh] views, no copies
    k = linear(k_in, W[d:2 * d], b[d:2 * d]).view(B, Lk, H, dh).permute(0, 2, 1, 3)
    v = linear(v_in, W[2 * d:], b[2 * d:]).view(B, Lk, H, dh).permute(0, 2, 1, 3)
    s = bmm4(q, k.transpose(2, 3), 1.0 / math.sqrt(dh))                                # [B, H, Lq, Lk]
    if kpm is not None:
        s = s.masked_fill(kpm.to(torch.bool).view(B, 1, 1, Lk), float('-inf'))
    p = torch.softmax(s, dim=-1)
    p = F.dropout(p, m.dropout, training)
    o = bmm4(p, v).permute(0, 2, 1, 3).reshape(B * Lq, d)
    return linear(o, m.out_proj.weight, m.out_proj.bias)


def decoder_forward2d(dec, t2, mem2, mem_pos2, B, Lq, S, tgt_kpm, mem_kpm):
    """TransformerDecoder (post-norm layers, detr_transformer.py:265-286, + the final norm :88) on batch-first rows."""
    for layer in dec.layers:
        t = layer.training
        a = mha(t2, t2, t2, layer.self_attn, B, Lq, Lq, tgt_kpm, t)
        t2 = layer_norm(t2 + F.dropout(a, layer.dropout1.p, t), layer.norm1)
        a = mha(t2, mem_pos2, mem2, layer.multihead_attn, B, Lq, S, mem_kpm, t)
        t2 = layer_norm(t2 + F.dropout(a, layer.dropout2.p, t), layer.norm2)
        h = F.dropout(linear(t2, layer.linear1.weight, layer.linear1.bias, relu=True), layer.dropout.p, t)
        t2 = layer_norm(t2 + F.dropout(linear(h, layer.linear2.weight, layer.linear2.bias), layer.dropout3.p, t), layer.norm3)
    if dec.norm is not None:
        t2 = layer_norm(t2, dec.norm)
    return t2
