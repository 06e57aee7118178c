"""`x1 = norm_a(x + dropout(r));  y = norm_b(x1 + dropout(linear2(dropout(relu(linear1(x1))))))` — the tail of every DETR encoder / decoder
layer (training/detr_transformer.py:210-214 / 280-285: the residual + LayerNorm that closes the attention sub-block, then the feed-forward
sub-block with its own residual + LayerNorm) as ONE autograd node with 3 launches forward and 4 backward (the unfused path: 4 and 6-7):

    forward   ldetr_layernorm_fwd_pos_f32 (x, r -> x1)
              ldetr_ffn_fwd_f32           (hidden slices across blocks, hidden tile kept on chip -> 32 partial outputs)
              ldetr_layernorm_fwd_parts_f32 (sums the partial outputs + linear2's bias, residual dropout, add x1, normalise [, + pos])
    backward  ldetr_layernorm_bwd2_f32     (norm_b: residual-path gradient dz, branch gradient dr)
              ldetr_ffn_bwd_f32            (hidden pre-activation gradient dh + 32 partial input gradients)
              ldetr_gemm_pair_f32 TN + TN  (dW2 += dr^T h, dW1 += dh^T x1, bias gradients as row sums, straight into the flat .grad)
              ldetr_layernorm_bwd_parts_f32 (norm_a: incoming gradient = dz + the 32 partial sums, in slice order -> dx, d r)

No atomics on the activation path: forward and input gradients are bit-reproducible run to run.  Used for token counts up to
MAX_ROWS (512: the decoder-side stacks, 144..320 tokens, where the two linear layers are latency-bound
launches); the image-token encoder (1024+ rows) keeps the large-tile GEMMs.
"""

import torch

from . import core

FUSED = True      # module switches (tests / A-B runs set them)
MAX_ROWS = core.knob('FFN_MAX_ROWS', 512)      # LDETR_DEBUG="FFN_MAX_ROWS=n": A/B of the fused block on the 64-token encoders (1024 / 2048 rows)


def usable(x2, linear1, linear2):
    D, F = linear1.weight.shape[1], linear1.weight.shape[0]
    return (FUSED and x2.is_cuda and D == 256 and F % 64 == 0 and x2.shape[0] <= MAX_ROWS and x2.dtype == torch.float32
            and linear1.bias is not None and linear2.bias is not None)


class _LnFfnLnFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, r, ga, ba, eps_a, p_a, w1, b1, w2, b2, gb, bb, eps_b, p_hidden, p_b, pos, r_bias=None):
        core.require_gpu(x, r, ga, ba, w1, b1, w2, b2, gb, bb, pos, r_bias)
        ctx.set_materialize_grads(False)
        D = x.shape[-1]
        # (a 3-D r with a 2-D x: per-head contributions of the fused self-attention block, summed with r_bias inside norm_a's launch)
        n_rparts = r.shape[0] if (r.dim() == 3 and x.dim() == 2) else 0
        x2 = core.f32c(x.reshape(-1, D))
        r2 = core.f32c(r) if n_rparts else core.f32c(r.reshape(-1, D))
        rb = core.f32c(r_bias) if (n_rparts and r_bias is not None) else None
        M, F = x2.shape[0], w1.shape[0]
        W1, B1, W2, B2 = core.f32c(w1.detach()), core.f32c(b1.detach()), core.f32c(w2.detach()), core.f32c(b2.detach())
        Ga, Ba, Gb, Bb = core.f32c(ga.detach()), core.f32c(ba.detach()), core.f32c(gb.detach()), core.f32c(bb.detach())
        new = lambda *shape: torch.empty(shape, device=x.device, dtype=torch.float32)
        # norm_a(x + dropout(r))
        x1, za, mean_a, rstd_a = new(M, D), new(M, D), new(M), new(M)
        seed_a = core.next_seed() if p_a > 0 else 0
        if n_rparts:
            core.check(core.lib().ldetr_layernorm_fwd_parts_f32(
                core.ptr(x2), core.ptr(r2), n_rparts, M * D, core.ptr(rb), core.ptr(Ga), core.ptr(Ba), core.ptr(x1), core.ptr(za), core.ptr(mean_a),
                core.ptr(rstd_a), M, D, eps_a, p_a, seed_a, core.seed_ptr() if p_a > 0 else None, None, 0, None, core.stream()), 'layernorm_fwd_parts')
        else:
            core.check(core.lib().ldetr_layernorm_fwd_pos_f32(
                core.ptr(x2), core.ptr(r2), core.ptr(Ga), core.ptr(Ba), core.ptr(x1), core.ptr(za), core.ptr(mean_a), core.ptr(rstd_a), M, D, eps_a, p_a,
                seed_a, core.seed_ptr() if p_a > 0 else None, None, 0, None, core.stream()), 'layernorm_fwd')
        # feed-forward block on x1, hidden slices across blocks
        ns = F // 64
        h, parts = new(M, F), new(ns, M, D)
        seed_h = core.next_seed() if p_hidden > 0 else 0
        core.check(core.lib().ldetr_ffn_fwd_f32(core.ptr(x1), D, core.ptr(W1), core.ptr(B1), core.ptr(W2), core.ptr(h), core.ptr(parts),
                                                M, D, F, p_hidden, seed_h, core.seed_ptr() if p_hidden > 0 else None, core.stream()), 'ffn_fwd')
        # norm_b(x1 + dropout(sum of the partial outputs + b2)) [, + pos]
        y, zb, mean_b, rstd_b = new(M, D), new(M, D), new(M), new(M)
        seed_b = core.next_seed() if p_b > 0 else 0
        pos2 = core.f32c(pos.reshape(-1, D)) if pos is not None else None
        ypos = new(M, D) if pos2 is not None else None
        core.check(core.lib().ldetr_layernorm_fwd_parts_f32(
            core.ptr(x1), core.ptr(parts), ns, M * D, core.ptr(B2), core.ptr(Gb), core.ptr(Bb), core.ptr(y), core.ptr(zb), core.ptr(mean_b), core.ptr(rstd_b),
            M, D, eps_b, p_b, seed_b, core.seed_ptr() if p_b > 0 else None, core.ptr(pos2), pos2.shape[0] if pos2 is not None else 0, core.ptr(ypos),
            core.stream()), 'layernorm_fwd_parts')
        ctx.save_for_backward(x1, h, W1, W2, za, mean_a, rstd_a, Ga, zb, mean_b, rstd_b, Gb)
        ctx.cfg = (x.shape, D, F, M, p_a, seed_a, p_hidden, p_b, seed_b)
        ctx.n_rparts = n_rparts
        ctx.params = (ga, ba, w1, b1, w2, b2, gb, bb)
        if pos is not None:
            return y.reshape(x.shape), ypos.reshape(x.shape)
        return y.reshape(x.shape)

    @staticmethod
    def backward(ctx, dy, dypos=None):
        x1, h, W1, W2, za, mean_a, rstd_a, Ga, zb, mean_b, rstd_b, Gb = ctx.saved_tensors
        xshape, D, F, M, p_a, seed_a, p_hidden, p_b, seed_b = ctx.cfg
        ga, ba, w1, b1, w2, b2, gb, bb = ctx.params
        nin = 17
        if dy is None and dypos is None:
            return (None,) * nin
        if dy is None:
            dy, dypos = dypos, None
        dev = dy.device
        new = lambda *shape: torch.empty(shape, device=dev, dtype=torch.float32)
        dy2 = core.f32c(dy.reshape(-1, D))
        dyp = core.f32c(dypos.reshape(-1, D)) if dypos is not None else None

        def ln_grads(gamma, beta, ig, ib):
            """-> (dgamma target, dbeta target, returned-to-autograd pair): the flat .grad views when available (the kernel's atomics accumulate)."""
            need = ctx.needs_input_grad[ig] or ctx.needs_input_grad[ib]
            fg, fb = core.flat_grad(gamma), core.flat_grad(beta)
            if need and fg is not None and fb is not None:
                return fg, fb, (None, None)
            if need:
                a, b = torch.zeros(D, device=dev, dtype=torch.float32), torch.zeros(D, device=dev, dtype=torch.float32)
                return a, b, (a, b)
            return None, None, (None, None)
        # norm_b backward: dz (gradient of its input sum = the residual-path gradient of x1) and dr (branch gradient, dropout mask applied)
        dgb, dbb, ret_b = ln_grads(gb, bb, 10, 11)
        dz, dr = new(M, D), new(M, D)
        core.check(core.lib().ldetr_layernorm_bwd2_f32(
            core.ptr(dy2), core.ptr(dyp), core.ptr(zb), core.ptr(mean_b), core.ptr(rstd_b), core.ptr(Gb), core.ptr(dz), core.ptr(dr),
            core.ptr(dgb), core.ptr(dbb), M, D, p_b, seed_b, core.seed_ptr() if p_b > 0 else None, core.stream()), 'layernorm_bwd')
        # feed-forward backward: partial input gradients per hidden slice (+ dh when the weights train)
        need_w = (ctx.needs_input_grad[6] or ctx.needs_input_grad[8]) and not core.WEIGHT_GRADIENTS_DISABLED[0]
        ns = F // 64
        dh = new(M, F) if need_w else None
        dxpart = new(ns, M, D)
        core.check(core.lib().ldetr_ffn_bwd_f32(core.ptr(dr), core.ptr(x1), D, core.ptr(h), core.ptr(W1), core.ptr(W2), core.ptr(dxpart), core.ptr(dh),
                                                M, D, F, p_hidden, core.stream()), 'ffn_bwd')
        out_w = [None] * 4
        if need_w:
            # dW2 += dr^T h and dW1 += dh^T x1 (K = tokens), bias gradients as the row sums of the transposed operands: one paired launch
            flat = [core.flat_grad(t) for t in (w1, b1, w2, b2)]
            if all(f is not None and f.is_contiguous() for f in flat):
                tgt = flat
            else:
                tgt = [torch.zeros_like(t, memory_format=torch.contiguous_format) for t in (w1, b1, w2, b2)]
                out_w = tgt
            core.gemm_pair(dict(A=dr, B=h, ta=1, tb=1, M=D, N=F, K=M, out=tgt[2], ep=core.epilogue(accumulate=True, a_rowsum=tgt[3])),
                           dict(A=dh, B=x1, ta=1, tb=1, M=F, N=D, K=M, out=tgt[0], ep=core.epilogue(accumulate=True, a_rowsum=tgt[1])))
        # norm_a backward: incoming gradient = dz + sum of the partial input gradients
        dga, dba, ret_a = ln_grads(ga, ba, 2, 3)
        need_x, need_r = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        dx = new(M, D)
        drr = new(M, D) if (need_r and p_a > 0) else None
        core.check(core.lib().ldetr_layernorm_bwd_parts_f32(
            core.ptr(dz), None, core.ptr(dxpart), ns, M * D, core.ptr(za), core.ptr(mean_a), core.ptr(rstd_a), core.ptr(Ga), core.ptr(dx), core.ptr(drr),
            core.ptr(dga), core.ptr(dba), M, D, p_a, seed_a, core.seed_ptr() if p_a > 0 else None, core.stream()), 'layernorm_bwd')
        g_x = dx.reshape(xshape) if need_x else None
        g_r = None
        if need_r:
            g_r = drr if drr is not None else dx
            g_r = g_r.unsqueeze(0).expand(ctx.n_rparts, M, D) if ctx.n_rparts else g_r.reshape(xshape)
        return (g_x, g_r, ret_a[0], ret_a[1], None, None, out_w[0], out_w[1], out_w[2], out_w[3], ret_b[0], ret_b[1], None, None, None, None, None)


def add_ln_ffn_add_ln(x, r, norm_a, p_a, linear1, linear2, norm_b, p_hidden=0.0, p_b=0.0, pos=None, r_bias=None):
    """norm_b(x1 + drop_b(linear2(drop_h(relu(linear1(x1)))))) with x1 = norm_a(x + drop_a(r)); pos as in hip.layernorm.add_layernorm
    (-> (y, y + pos))."""
    return _LnFfnLnFn.apply(x, r, norm_a.weight, norm_a.bias, norm_a.eps, p_a, linear1.weight, linear1.bias, linear2.weight, linear2.bias,
                            norm_b.weight, norm_b.bias, norm_b.eps, p_hidden, p_b, pos, r_bias)


class _FfnLargeFn(torch.autograd.Function):
    """linear2(dropout(relu(linear1(x)))) above the fused launch's token limit (the 64-token encoders: 1024+ rows) as ONE autograd node around the
    engine's GEMMs: forward = the two launches of two hip.linear nodes (same epilogues, same dropout element index); backward = 3 launches instead
    of 5: the hidden gradient dH = (dY W2) * relu'(h) / keep comes out of the first GEMM's epilogue (mask from the saved hidden activation: a dropped
    or rectified element is 0 there), so no activation-gradient pass reads and rewrites it; both weight gradients (+ bias gradients as row sums of
    their transposed operands) are one paired launch; dX = dH W1 takes the residual-path gradient in its epilogue.  Returns (y, alias of x): the
    residual branch reads x through the alias, so this node is x's only consumer (hip.linear._LinearFn.forward)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, p_hidden):
        core.require_gpu(x, w1, b1, w2, b2)
        ctx.set_materialize_grads(False)
        D, F = w1.shape[1], w1.shape[0]
        x2 = core.f32c(x.reshape(-1, D))
        W1, B1, W2, B2 = core.f32c(w1.detach()), core.f32c(b1.detach()), core.f32c(w2.detach()), core.f32c(b2.detach())
        M = x2.shape[0]
        seed = core.next_seed() if p_hidden > 0 else 0
        h = core.gemm(x2, W1, 0, 0, M, F, D, ep=core.epilogue(col_bias=B1, act=core.ACT_RELU, p_drop=p_hidden, seed=seed))
        y = core.gemm(h, W2, 0, 0, M, D, F, ep=core.epilogue(col_bias=B2))
        ctx.save_for_backward(x2, h, W1, W2)
        ctx.cfg = (x.shape, D, F, M, p_hidden)
        ctx.params = (w1, b1, w2, b2)
        return y.reshape(x.shape), x

    @staticmethod
    def backward(ctx, dy, dx_pass=None):
        x2, h, W1, W2 = ctx.saved_tensors
        xshape, D, F, M, p_hidden = ctx.cfg
        w1, b1, w2, b2 = ctx.params
        if dy is None:
            return (dx_pass,) + (None,) * 5
        dy2 = core.f32c(dy.reshape(-1, D))
        need_x = ctx.needs_input_grad[0]
        need_w = any(ctx.needs_input_grad[1:5]) and not core.WEIGHT_GRADIENTS_DISABLED[0]
        # dH = (dY W2) masked by the saved hidden activation, 1 / keep folded into alpha
        dh = core.gemm(dy2, W2, 0, 1, M, F, D, ep=core.epilogue(alpha=1.0 / (1.0 - p_hidden) if p_hidden > 0 else 1.0, mask_src=h, mask_mode=1))
        out_w = [None] * 4
        if need_w:
            flat = [core.flat_grad(t) for t in (w1, b1, w2, b2)]
            if all(f is not None and f.is_contiguous() for f in flat):
                tgt = flat
            else:
                tgt = [torch.zeros_like(t, memory_format=torch.contiguous_format) for t in (w1, b1, w2, b2)]
                out_w = tgt
            core.gemm_pair(dict(A=dy2, B=h, ta=1, tb=1, M=D, N=F, K=M, out=tgt[2], ep=core.epilogue(accumulate=True, a_rowsum=tgt[3])),
                           dict(A=dh, B=x2, ta=1, tb=1, M=F, N=D, K=M, out=tgt[0], ep=core.epilogue(accumulate=True, a_rowsum=tgt[1])))
        dx = None
        if need_x:
            res = core.f32c(dx_pass.reshape(-1, D)) if dx_pass is not None else None
            dx = core.gemm(dh, W1, 0, 1, M, D, F, ep=core.epilogue(residual=res)).reshape(xshape)
        elif dx_pass is not None:
            dx = dx_pass
        return dx, out_w[0], out_w[1], out_w[2], out_w[3], None


def large_usable(x2, linear1, linear2):
    return (FUSED and x2.is_cuda and x2.dtype == torch.float32 and linear1.bias is not None and linear2.bias is not None
            and linear1.weight.shape[0] % 4 == 0 and linear1.weight.shape[1] % 4 == 0)


def ffn_large(x, linear1, linear2, p_hidden=0.0):
    """-> (linear2(dropout(relu(linear1(x)))), alias of x for the residual branch)."""
    return _FfnLargeFn.apply(x, linear1.weight, linear1.bias, linear2.weight, linear2.bias, p_hidden)
