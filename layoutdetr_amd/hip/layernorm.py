"""y = LayerNorm(x + dropout(residual)) fused into one forward and one backward launch.
Reference chain: training/detr_transformer.py:210-214 / 275-285 (post-norm blocks, eps 1e-5)."""
import torch

from . import core


class _AddLnFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, r, gamma, beta, eps, p_drop, pos=None, r_bias=None):
        core.require_gpu(x, r, gamma, beta, pos, r_bias)
        ctx.set_materialize_grads(False)
        D = x.shape[-1]
        x2 = core.f32c(x.reshape(-1, D))
        # a 3-D residual [S, rows, D] is a sum still to be formed: r = r_bias + sum_s r[s] (per-head / per-slice contributions of a fused
        # sub-block), added in slice order inside this launch
        n_parts = r.shape[0] if (r is not None and r.dim() == 3 and x.dim() == 2) else 0
        r2 = (core.f32c(r) if n_parts else core.f32c(r.reshape(-1, D))) if r is not None else None
        rb = core.f32c(r_bias) if (n_parts and r_bias is not None) else None
        g, b = core.f32c(gamma), core.f32c(beta)
        rows = x2.shape[0]
        y = torch.empty_like(x2)
        z = torch.empty_like(x2) if r2 is not None else x2
        mean = torch.empty(rows, device=x.device, dtype=torch.float32)
        rstd = torch.empty(rows, device=x.device, dtype=torch.float32)
        seed = core.next_seed() if (p_drop > 0 and r2 is not None) else 0
        pos2 = core.f32c(pos.reshape(-1, D)) if pos is not None else None
        ypos = torch.empty_like(x2) if pos2 is not None else None
        if n_parts:
            core.check(core.lib().ldetr_layernorm_fwd_parts_f32(
                core.ptr(x2), core.ptr(r2), n_parts, rows * D, core.ptr(rb), core.ptr(g), core.ptr(b), core.ptr(y), core.ptr(z),
                core.ptr(mean), core.ptr(rstd), rows, D, eps, p_drop, seed, core.seed_ptr() if seed else None,
                core.ptr(pos2), pos2.shape[0] if pos2 is not None else 0, core.ptr(ypos), core.stream()), 'layernorm_fwd_parts')
        else:
            core.check(core.lib().ldetr_layernorm_fwd_pos_f32(
                core.ptr(x2), core.ptr(r2), core.ptr(g), core.ptr(b), core.ptr(y), core.ptr(z) if r2 is not None else None,
                core.ptr(mean), core.ptr(rstd), rows, D, eps, p_drop if r2 is not None else 0.0, seed,
                core.seed_ptr() if seed else None, core.ptr(pos2), pos2.shape[0] if pos2 is not None else 0, core.ptr(ypos), core.stream()),
                'layernorm_fwd')
        ctx.save_for_backward(z, mean, rstd, g)
        ctx.n_parts = n_parts
        ctx.cfg = (x.shape, r is not None, p_drop if r2 is not None else 0.0, seed, D)
        ctx.params = (gamma, beta)
        if pos is not None:
            return y.reshape(x.shape), ypos.reshape(x.shape)
        return y.reshape(x.shape)

    @staticmethod
    def backward(ctx, dy, dypos=None):
        z, mean, rstd, g = ctx.saved_tensors
        xshape, has_r, p_drop, seed, D = ctx.cfg
        if dy is None and dypos is None:
            return (None,) * 8
        if dy is None:
            dy, dypos = dypos, None
        dy2 = core.f32c(dy.reshape(-1, D))
        dyp = core.f32c(dypos.reshape(-1, D)) if dypos is not None else None
        rows = dy2.shape[0]
        need_x, need_r = ctx.needs_input_grad[0], has_r and ctx.needs_input_grad[1]
        need_g = ctx.needs_input_grad[2] or ctx.needs_input_grad[3]
        dx = torch.empty_like(dy2)
        dr = None
        if need_r:
            dr = torch.empty_like(dy2) if p_drop > 0 else dx
        gg, gbt = core.flat_grad(ctx.params[0]), core.flat_grad(ctx.params[1])
        fused = need_g and gg is not None and gbt is not None
        if fused:
            dgamma, dbeta = gg, gbt          # the kernel's atomics accumulate straight into the flat .grad views
        else:
            dgamma = torch.zeros(D, device=dy2.device, dtype=torch.float32) if need_g else None
            dbeta = torch.zeros(D, device=dy2.device, dtype=torch.float32) if need_g else None
        core.check(core.lib().ldetr_layernorm_bwd2_f32(
            core.ptr(dy2), core.ptr(dyp), core.ptr(z), core.ptr(mean), core.ptr(rstd), core.ptr(g), core.ptr(dx),
            core.ptr(dr) if (need_r and p_drop > 0) else None, core.ptr(dgamma), core.ptr(dbeta), rows, D, p_drop, seed,
            core.seed_ptr() if p_drop > 0 else None, core.stream()), 'layernorm_bwd')
        if fused:
            dgamma = dbeta = None
        if need_r:
            g_r = dr.unsqueeze(0).expand(ctx.n_parts, rows, D) if ctx.n_parts else dr.reshape(xshape)     # (every slice of a partial-sum residual receives the same gradient)
        else:
            g_r = None
        return (dx.reshape(xshape) if need_x else None, g_r, dgamma, dbeta, None, None, None, None)


def add_layernorm(x, residual, gamma, beta, eps=1e-5, p_drop=0.0, pos=None, r_bias=None):
    """LayerNorm(x + dropout(residual)); residual may be None (plain LayerNorm).
    residual [S, rows, D] with x [rows, D]: the residual is r_bias + the sum of the S slices (formed inside the launch, in slice order).
    pos ([S, D], rows broadcast over the batch): returns (y, y + pos) — the second tensor is what the next attention block projects
    q / k from, produced by the same launch instead of a separate add; its gradient is summed inside the backward launch."""
    return _AddLnFn.apply(x, residual, gamma, beta, eps, p_drop, pos, r_bias)
