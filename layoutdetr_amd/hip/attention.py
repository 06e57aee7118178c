"""Fused attention core + multi-head-attention block on the HIP kernels.

`mha_forward` reproduces nn.MultiheadAttention(d, h, dropout) as the reference calls it
(training/detr_transformer.py:208-209, 273-274, 277-280): packed in_proj_weight [3d, d], scale
1/sqrt(dh) on q, boolean key_padding_mask -> -inf, dropout on the probabilities, out_proj; the
head-averaged attention weights the reference discards (`[0]`) are never formed.
Layout is batch-first: activations are [B*L, d] row-major (row = b*L + position).
"""
import math

import torch

from . import core
from .linear import linear


class _AttnFn(torch.autograd.Function):
    """q: [B*Lq, >=d] view, k/v: [B*Lk, >=d] views (unit inner stride, arbitrary row stride)."""

    @staticmethod
    def forward(ctx, q, k, v, kpm, B, H, Lq, Lk, p_drop):
        core.require_gpu(q, k, v, kpm)
        dh = q.shape[1] // H
        assert q.stride(1) == 1 and k.stride(1) == 1 and v.stride(1) == 1
        d = H * dh
        out = torch.empty((B * Lq, d), device=q.device, dtype=torch.float32)
        lse = torch.empty((B * H * Lq,), device=q.device, dtype=torch.float32)
        seed = core.next_seed() if p_drop > 0 else 0
        scale = 1.0 / math.sqrt(dh)
        core.check(core.lib().ldetr_attention_fwd_f32(
            core.ptr(q), q.stride(0), core.ptr(k), k.stride(0), core.ptr(v), v.stride(0), core.ptr(kpm),
            core.ptr(out), d, core.ptr(lse), B, H, Lq, Lk, dh, scale, p_drop, seed, core.seed_ptr() if p_drop > 0 else None,
            0, core.stream()), 'attention_fwd')
        ctx.save_for_backward(q, k, v, kpm, out, lse)
        ctx.cfg = (B, H, Lq, Lk, dh, scale, p_drop, seed)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, kpm, out, lse = ctx.saved_tensors
        B, H, Lq, Lk, dh, scale, p_drop, seed = ctx.cfg
        d = H * dh
        dout = core.f32c(dout)
        dq = torch.empty((B * Lq, d), device=q.device, dtype=torch.float32)
        dk = torch.empty((B * Lk, d), device=q.device, dtype=torch.float32)
        dv = torch.empty((B * Lk, d), device=q.device, dtype=torch.float32)
        core.check(core.lib().ldetr_attention_bwd_f32(
            core.ptr(q), q.stride(0), core.ptr(k), k.stride(0), core.ptr(v), v.stride(0), core.ptr(kpm),
            core.ptr(out), d, core.ptr(lse), core.ptr(dout), d, core.ptr(dq), d, core.ptr(dk), d, core.ptr(dv), d,
            B, H, Lq, Lk, dh, scale, p_drop, seed, core.seed_ptr() if p_drop > 0 else None, 0, core.stream()), 'attention_bwd')
        return dq, dk, dv, None, None, None, None, None, None


class _AttnPackedFn(torch.autograd.Function):
    """Self-attention on a packed projection: qkv is [B*L, 3d] (v_sep None) or [B*L, 2d] = (q | k) with v separate.
    Autograd sees ONE input and gets ONE packed gradient: slicing q/k/v out of the projection as autograd views costs
    a zero-fill + copy per slice and two adds per attention in backward (SliceBackward), 8 launches of pure overhead."""

    @staticmethod
    def forward(ctx, qkv, v_sep, kpm, B, H, L, p_drop, causal=False):
        core.require_gpu(qkv, v_sep, kpm)
        assert qkv.stride(1) == 1
        d = qkv.shape[1] // (3 if v_sep is None else 2)
        dh = d // H
        q, k = qkv[:, :d], qkv[:, d:2 * d]
        v = qkv[:, 2 * d:] if v_sep is None else v_sep
        assert v.stride(1) == 1
        out = torch.empty((B * L, d), device=qkv.device, dtype=torch.float32)
        lse = torch.empty((B * H * L,), device=qkv.device, dtype=torch.float32)
        seed = core.next_seed() if p_drop > 0 else 0
        scale = 1.0 / math.sqrt(dh)
        core.check(core.lib().ldetr_attention_fwd_f32(
            core.ptr(q), q.stride(0), core.ptr(k), k.stride(0), core.ptr(v), v.stride(0), core.ptr(kpm),
            core.ptr(out), d, core.ptr(lse), B, H, L, L, dh, scale, p_drop, seed, core.seed_ptr() if p_drop > 0 else None,
            1 if causal else 0, core.stream()), 'attention_fwd')
        ctx.save_for_backward(qkv, v_sep, kpm, out, lse)
        ctx.cfg = (B, H, L, dh, scale, p_drop, seed, 1 if causal else 0)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, v_sep, kpm, out, lse = ctx.saved_tensors
        B, H, L, dh, scale, p_drop, seed, causal = ctx.cfg
        d = H * dh
        q, k = qkv[:, :d], qkv[:, d:2 * d]
        v = qkv[:, 2 * d:] if v_sep is None else v_sep
        dout = core.f32c(dout)
        dqkv = torch.empty_like(qkv)
        dq, dk = dqkv[:, :d], dqkv[:, d:2 * d]
        dv = dqkv[:, 2 * d:] if v_sep is None else torch.empty((B * L, d), device=qkv.device, dtype=torch.float32)
        core.check(core.lib().ldetr_attention_bwd_f32(
            core.ptr(q), q.stride(0), core.ptr(k), k.stride(0), core.ptr(v), v.stride(0), core.ptr(kpm),
            core.ptr(out), d, core.ptr(lse), core.ptr(dout), d, core.ptr(dq), dq.stride(0), core.ptr(dk), dk.stride(0),
            core.ptr(dv), dv.stride(0), B, H, L, L, dh, scale, p_drop, seed, core.seed_ptr() if p_drop > 0 else None,
            causal, core.stream()), 'attention_bwd')
        return dqkv, (None if v_sep is None else dv), None, None, None, None, None, None


def _kpm_u8(key_padding_mask):
    return None if key_padding_mask is None else key_padding_mask.to(torch.uint8).contiguous()


def attention(q, k, v, key_padding_mask, B, H, Lq, Lk, p_drop=0.0):
    kpm = None
    if key_padding_mask is not None:
        kpm = key_padding_mask.to(torch.uint8).contiguous()
    return _AttnFn.apply(q, k, v, kpm, B, H, Lq, Lk, p_drop)


def mha_forward(query, key, value, in_proj_weight, in_proj_bias, out_w, out_b, nhead, B, Lq, Lk,
                key_padding_mask=None, p_drop=0.0, same_qk=False, same_qkv=False, qk_pos=None, passthru=False, qk_in=None, kv_alias=False):
    """query: [B*Lq, d]; key/value: [B*Lk, d].  Returns [B*Lq, d] (before the caller's residual/dropout/LN).

    same_qkv: query, key and value are one tensor  -> one packed [3d] projection (decoder self-attention).
    same_qk:  query and key are `query (+ qk_pos)`, value is `value` (= query without the position term: encoder self-attention)
              -> one [2d] projection + one [d] projection; qk_pos is added inside the projection node.
    passthru: also return an alias of `query` that already went through the projection nodes; feeding the residual branch
              from it keeps autograd from summing the gradients of `query` with separate add launches (hip/linear.py).
    qk_in:    (same_qk) the tensor `query + qk_pos` already formed by the producer (layernorm's second output): q / k are projected
              from it with no add launch here; its gradient flows back to that producer.
    kv_alias: (cross-attention) also return aliases of `key` and `value` that went through their projection nodes: the next decoder
              layer reads the memory through them, so the memory's gradient accumulates inside the projection GEMMs' epilogues
              instead of 2 x (layers - 1) autograd add launches per decoder stack.
    """
    d = query.shape[1]
    W, bvec = in_proj_weight, in_proj_bias
    alias = query
    if same_qkv:
        r = linear(query, W, bvec, passthru=passthru)
        qkv, alias = r if passthru else (r, query)
        o = _AttnPackedFn.apply(qkv, None, _kpm_u8(key_padding_mask), B, nhead, Lq, p_drop)
    elif same_qk and qk_in is not None:
        qk = linear(qk_in, W, bvec, rows=(0, 2 * d))
        r = linear(query, W, bvec, rows=(2 * d, 3 * d), passthru=passthru)
        v, alias = r if passthru else (r, query)
        o = _AttnPackedFn.apply(qk, v, _kpm_u8(key_padding_mask), B, nhead, Lq, p_drop)
    elif same_qk:
        r = linear(query, W, bvec, rows=(0, 2 * d), add_input=qk_pos, passthru=passthru)
        qk, alias = r if passthru else (r, query)
        vin = alias if value is query else value
        r = linear(vin, W, bvec, rows=(2 * d, 3 * d), passthru=passthru and value is query)
        v, alias = r if (passthru and value is query) else (r, alias)
        o = _AttnPackedFn.apply(qk, v, _kpm_u8(key_padding_mask), B, nhead, Lq, p_drop)
    else:
        r = linear(query, W, bvec, rows=(0, d), passthru=passthru)
        q, alias = r if passthru else (r, query)
        if kv_alias:
            k, key = linear(key, W, bvec, rows=(d, 2 * d), passthru=True)
            v, value = linear(value, W, bvec, rows=(2 * d, 3 * d), passthru=True)
        else:
            k = linear(key, W, bvec, rows=(d, 2 * d))
            v = linear(value, W, bvec, rows=(2 * d, 3 * d))
        o = attention(q, k, v, key_padding_mask, B, nhead, Lq, Lk, p_drop)
    out = linear(o, out_w, out_b)
    if kv_alias:
        return out, alias, key, value
    return (out, alias) if passthru else out
