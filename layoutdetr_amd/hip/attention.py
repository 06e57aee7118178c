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
    def forward(ctx, q, k, v, kpm, B, H, Lq, Lk, p_drop, kv_grad_dst=None):
        core.require_gpu(q, k, v, kpm)
        ctx.kv_grad_dst = kv_grad_dst      # callable -> (dk view, dv view) inside a grouped projection's gradient buffer (grouped_kv below)
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
        if ctx.kv_grad_dst is not None:
            dk, dv = ctx.kv_grad_dst()       # written in place: the grouped projection's backward reads the whole buffer
        else:
            dk = torch.empty((B * Lk, d), device=q.device, dtype=torch.float32)
            dv = torch.empty((B * Lk, d), device=q.device, dtype=torch.float32)
        core.check(core.lib().ldetr_attention_bwd_f32(
            core.ptr(q), q.stride(0), core.ptr(k), k.stride(0), core.ptr(v), v.stride(0), core.ptr(kpm),
            core.ptr(out), d, core.ptr(lse), core.ptr(dout), d, core.ptr(dq), d, core.ptr(dk), dk.stride(0), core.ptr(dv), dv.stride(0),
            B, H, Lq, Lk, dh, scale, p_drop, seed, core.seed_ptr() if p_drop > 0 else None, 0, core.stream()), 'attention_bwd')
        return dq, dk, dv, None, None, None, None, None, None, None


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
    """bool / uint8 key-padding mask -> contiguous uint8 (a contiguous bool mask is reinterpreted in place: one byte per element holding 0 / 1)."""
    m = key_padding_mask
    if m is None:
        return None
    if m.dtype == torch.bool:
        return (m if m.is_contiguous() else m.contiguous()).view(torch.uint8)
    return m.to(torch.uint8).contiguous()


def attention(q, k, v, key_padding_mask, B, H, Lq, Lk, p_drop=0.0, kv_grad_dst=None):
    return _AttnFn.apply(q, k, v, _kpm_u8(key_padding_mask), B, H, Lq, Lk, p_drop, kv_grad_dst)


class _GroupedKVFn(torch.autograd.Function):
    """The memory K / V projections of ALL decoder layers as two GEMMs.  The reference projects the (layer-invariant) encoder memory
    inside every decoder layer (detr_transformer.py:277-280: `multihead_attn(query, key=memory + pos, value=memory)`, each layer with
    its own in_proj_weight rows d:2d / 2d:3d): 2 x layers projections of M = S*B rows with N = d.  Here: K_all = (memory + pos) Wk_all^T
    and V_all = memory Wv_all^T with N = layers * d (one large-tile launch each instead of `layers` latency-bound ones), the layers
    read their [M, d] column blocks in place, their attention backward writes dK / dV into the matching blocks of ONE gradient
    buffer, and the backward is two data-gradient GEMMs (K = layers * d) + two weight-gradient GEMMs whose [layers, d, d] results are
    added into the layers' flat .grad rows with one strided launch.  The memory and memory + pos tensors are consumed once each,
    so autograd has no gradient fan-in to sum.
    forward(mem_pos, mem, n, W_0, b_0, ..., W_{n-1}, b_{n-1}) -> (K_0, ..., K_{n-1}, V_0, ..., V_{n-1}) and the gradient-buffer views via
    ctx (see grouped_kv)."""

    @staticmethod
    def forward(ctx, mem_pos, mem, holder, *params):
        core.require_gpu(mem_pos, mem)
        ctx.set_materialize_grads(False)
        Ws, bs = params[0::2], params[1::2]
        n, d = len(Ws), mem.shape[1]
        M = mem.shape[0]
        mp, mm = core.f32c(mem_pos), core.f32c(mem)
        with torch.no_grad():
            Wk = torch.cat([w.detach()[d:2 * d] for w in Ws] + [w.detach()[2 * d:] for w in Ws])        # [2 n d, d]: K rows of every layer, then V rows
            bk = torch.cat([b.detach()[d:2 * d] for b in bs] + [b.detach()[2 * d:] for b in bs])        # [2 n d]
        nd = n * d
        KV = torch.empty((M, 2 * nd), device=mem.device, dtype=torch.float32)
        core.gemm(mp, Wk[:nd], 0, 0, M, nd, d, out=KV[:, :nd], ep=core.epilogue(col_bias=bk[:nd]))
        core.gemm(mm, Wk[nd:], 0, 0, M, nd, d, out=KV[:, nd:], ep=core.epilogue(col_bias=bk[nd:]))
        ctx.save_for_backward(mp, mm, Wk)
        ctx.params = (Ws, bs)
        ctx.cfg = (n, d, M)
        ctx.holder = holder
        holder.dK = holder.dV = None
        return tuple(KV[:, i * d:(i + 1) * d] for i in range(2 * n))

    @staticmethod
    def backward(ctx, *grads):
        mp, mm, Wk = ctx.saved_tensors
        n, d, M = ctx.cfg
        Ws, bs = ctx.params
        nd = n * d
        dK, dV = ctx.holder.dK, ctx.holder.dV          # two packed [M, n d] buffers (the weight-gradient GEMM's row sums need a packed operand)
        fresh = dK is None
        if fresh:
            dK = core.zeros((M, nd), mp.device); dV = core.zeros((M, nd), mp.device)
        for i, g in enumerate(grads):      # blocks the attention backward wrote in place arrive as views of dK / dV: nothing to do for them
            dst = (dK if i < n else dV)[:, (i % n) * d:(i % n + 1) * d]
            if g is None:
                if not fresh:
                    dst.zero_()
            elif g.data_ptr() != dst.data_ptr() or g.stride() != dst.stride():
                dst.copy_(g)
        ctx.holder.dK = ctx.holder.dV = None
        dmp = core.gemm(dK, Wk[:nd], 0, 1, M, d, nd) if ctx.needs_input_grad[0] else None
        dmm = core.gemm(dV, Wk[nd:], 0, 1, M, d, nd) if ctx.needs_input_grad[1] else None
        need_w = any(ctx.needs_input_grad[3 + 2 * i] for i in range(n)) and not core.WEIGHT_GRADIENTS_DISABLED[0]
        out = [None] * (2 * n)
        if need_w:
            T = torch.empty((2, nd, d), device=mp.device, dtype=torch.float32)
            tb = core.zeros((2, nd), mp.device)
            core.gemm(dK, mp, 1, 1, nd, d, M, out=T[0], ep=core.epilogue(a_rowsum=tb[0]))
            core.gemm(dV, mm, 1, 1, nd, d, M, out=T[1], ep=core.epilogue(a_rowsum=tb[1]))
            gws, gbs = [core.flat_grad(w) for w in Ws], [core.flat_grad(b) for b in bs]
            strided = all(g is not None for g in gws + gbs) and n > 1
            if strided:       # the layers' parameters sit at a constant pitch in the flat gradient buffer: one strided add for all of them
                sw = gws[1].storage_offset() - gws[0].storage_offset(); sb = gbs[1].storage_offset() - gbs[0].storage_offset()
                strided = (all(g.is_contiguous() and g.untyped_storage().data_ptr() == gws[0].untyped_storage().data_ptr() for g in gws + gbs)
                           and all(gws[i].storage_offset() - gws[0].storage_offset() == i * sw and gbs[i].storage_offset() - gbs[0].storage_offset() == i * sb for i in range(n)))
            if strided:
                gw = torch.as_strided(gws[0], (n, 2, d, d), (sw, d * d, d, 1), gws[0].storage_offset() + d * d)
                gw += T.view(2, n, d, d).permute(1, 0, 2, 3)
                gb = torch.as_strided(gbs[0], (n, 2, d), (sb, d, 1), gbs[0].storage_offset() + d)
                gb += tb.view(2, n, d).permute(1, 0, 2)
            else:
                for i in range(n):
                    if gws[i] is not None:
                        gws[i][d:2 * d] += T[0, i * d:(i + 1) * d]; gws[i][2 * d:] += T[1, i * d:(i + 1) * d]
                    else:
                        g = torch.zeros_like(Ws[i]); g[d:2 * d] = T[0, i * d:(i + 1) * d]; g[2 * d:] = T[1, i * d:(i + 1) * d]
                        out[2 * i] = g
                    if gbs[i] is not None:
                        gbs[i][d:2 * d] += tb[0, i * d:(i + 1) * d]; gbs[i][2 * d:] += tb[1, i * d:(i + 1) * d]
                    else:
                        g = torch.zeros_like(bs[i]); g[d:2 * d] = tb[0, i * d:(i + 1) * d]; g[2 * d:] = tb[1, i * d:(i + 1) * d]
                        out[2 * i + 1] = g
        return (dmp, dmm, None) + tuple(out)


class _KVHolder(object):
    """Shared between a grouped projection and the attention calls that read it: the two [M, n d] gradient buffers (allocated by the first
    attention backward that needs them)."""
    dK = dV = None


def grouped_kv(mem_pos, mem, mhas):
    """mhas: the layers' nn.MultiheadAttention parameter containers -> [(K_i, V_i, grad_dst_i)] per layer; grad_dst_i() yields the (dK, dV)
    views the layer's attention backward writes into (pass it to mha_cross_kv)."""
    n, d, M = len(mhas), mem.shape[1], mem.shape[0]
    holder = _KVHolder()
    outs = _GroupedKVFn.apply(mem_pos, mem, holder, *[t for m in mhas for t in (m.in_proj_weight, m.in_proj_bias)])

    def dst(i):
        def views():
            if holder.dK is None:      # zero-filled: a layer whose attention backward never runs contributes nothing
                holder.dK = core.zeros((M, n * d), mem.device)
                holder.dV = core.zeros((M, n * d), mem.device)
            return holder.dK[:, i * d:(i + 1) * d], holder.dV[:, i * d:(i + 1) * d]
        return views
    return [(outs[i], outs[n + i], dst(i)) for i in range(n)]


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


def mha_cross_kv(query, K, V, grad_dst, in_proj_weight, in_proj_bias, out_w, out_b, nhead, B, Lq, Lk, key_padding_mask=None, p_drop=0.0):
    """Cross-attention block whose key / value projections were made by grouped_kv(): q projection (rows 0:d of the packed weight),
    fused attention on the layer's [M, d] blocks of the grouped buffer, output projection.  -> (block output, alias of `query` for the residual
    branch).  (The <= 16-query stacks at d_model 256 run inside hip.stacks instead: one launch per direction for the whole sub-block.)"""
    d = query.shape[1]
    q, alias = linear(query, in_proj_weight, in_proj_bias, rows=(0, d), passthru=True)
    o = attention(q, K, V, key_padding_mask, B, nhead, Lq, Lk, p_drop, kv_grad_dst=grad_dst)
    return linear(o, out_w, out_b), alias
