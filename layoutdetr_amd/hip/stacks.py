"""The short token stacks of LayoutDETR -- G's / D's layout decoders (training/detr_transformer.py:265-286 x 6, 9-10 queries per sample), D's
unconditional encoder and D's two reconstruction decoders (nn.TransformerEncoderLayer x 6: training/util.py:13-43, networks_detr.py:242-243, 269,
275-276) -- as ONE autograd node per group of up to two independent stacks.

Why a node per GROUP of STACKS and not per sub-block (rounds 3-4: hip.attention._SelfAttnPartsFn / _CrossAttnPartsFn, hip.ffn._LnFfnLnFn):
  * every launch of such a stack is a fraction of a wave of work per CU, so two stacks that do not depend on each other (D's conditional and
    unconditional reconstruction decoders; D's layout decoder and its unconditional encoder) advance in lock-step with ONE launch per sub-block step:
    the second problem's blocks follow the first's in the same grid (csrc/{mha_small,ffn_fused,layernorm}.hip take one or two argument blocks);
  * inside the node gradients travel between sub-blocks as PARTIAL SUMS: the one-launch attention backward (ldetr_mha_small_bwd_group_f32 /
    ldetr_mha_cross_bwd_f32: output-projection data gradient + attention backward + input-projection data gradient, three launches of the unfused
    path) leaves one input-gradient slice per head, which the LayerNorm backward in front adds in head order while it loads its incoming gradient --
    autograd would need a materialised tensor (one more launch) at every node boundary;
  * the weight gradients of a layer (feed-forward W1 / W2, attention in_proj / out_proj: independent contractions over the tokens) are ONE launch
    (ldetr_wgrad_multi_f32) instead of three paired ones.
Per layer: forward 4 launches (decoder: 6), backward 5 (decoder: 7) -- for one stack or for two.  No atomics on the activation path: forward values and
input gradients are bit-reproducible and identical whether a stack runs alone or in a group.
"""
import math

import torch

from .. import _lib
from . import core

D_MODEL, N_HEAD, MAX_TOKENS, MAX_ROWS, MAX_CROSS_KEYS = 256, 8, 16, 512, 64
ENABLED = core.knob('TOKEN_STACKS', 1) != 0    # 0: every sub-block as its own autograd node on the generic kernels (A/B and equivalence tests)

NODE_RUNS = [0]   # how many stack nodes ran (tests assert that a stack took this path)

ENC_PARAMS = 12   # self_attn.in_proj_weight, .in_proj_bias, .out_proj.weight, .out_proj.bias, norm1.w, norm1.b, linear1.w, linear1.b, linear2.w, linear2.b, norm2.w, norm2.b
DEC_PARAMS = 18   # self_attn (4), norm1 (2), multihead_attn (4), norm2 (2), linear1 (2), linear2 (2), norm3 (2)


class Prog(object):
    """One stack: `layers` (TransformerEncoderLayer / TransformerDecoderLayer modules) applied to x [B*L, 256] (row = b * L + l).
    kind 'enc': x = norm1(x + SA(x)); x = norm2(x + FFN(x)).   kind 'dec': ... + cross-attention onto the projected memory kvs[i] = (K_i, V_i, grad_dst_i)
    (hip.attention.grouped_kv) of S tokens per sample between the two.  kpm / mem_kpm: uint8 key-padding masks [B, L] / [B, S] or None."""

    def __init__(self, kind, layers, x, B, L, kpm, training, final_norm=None, kvs=None, S=0, mem_kpm=None):
        self.kind, self.layers, self.x, self.B, self.L, self.kpm, self.training = kind, list(layers), x, B, L, kpm, training
        self.final_norm, self.kvs, self.S, self.mem_kpm = final_norm, kvs, S, mem_kpm


def layer_params(kind, layer):
    sa = layer.self_attn
    p = [sa.in_proj_weight, sa.in_proj_bias, sa.out_proj.weight, sa.out_proj.bias, layer.norm1.weight, layer.norm1.bias]
    if kind == 'dec':
        ca = layer.multihead_attn
        p += [ca.in_proj_weight, ca.in_proj_bias, ca.out_proj.weight, ca.out_proj.bias, layer.norm2.weight, layer.norm2.bias]
        p += [layer.linear1.weight, layer.linear1.bias, layer.linear2.weight, layer.linear2.bias, layer.norm3.weight, layer.norm3.bias]
    else:
        p += [layer.linear1.weight, layer.linear1.bias, layer.linear2.weight, layer.linear2.bias, layer.norm2.weight, layer.norm2.bias]
    return p


def usable(prog):
    """d_model 256 with 8 heads, at most 16 tokens per sample and 512 rows, hidden width a multiple of 64, fp32 rows the kernels can address."""
    x = prog.x
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.shape[1] == D_MODEL and 1 <= prog.L <= MAX_TOKENS and x.shape[0] == prog.B * prog.L
            and x.shape[0] <= MAX_ROWS and len(prog.layers) >= 1):
        return False
    for l in prog.layers:
        sa = l.self_attn
        if sa.num_heads != N_HEAD or sa.in_proj_weight.shape[1] != D_MODEL or l.linear1.weight.shape[0] % 64 != 0 or l.linear1.bias is None or l.linear2.bias is None:
            return False
        if sa.in_proj_bias is None or sa.out_proj.bias is None:        # the group kernels read both unconditionally (nn.MultiheadAttention(bias=False) takes the generic path)
            return False
        if prog.kind == 'dec' and (l.multihead_attn.num_heads != N_HEAD or l.multihead_attn.in_proj_bias is None or l.multihead_attn.out_proj.bias is None):
            return False
    if prog.kind == 'dec' and (prog.kvs is None or len(prog.kvs) != len(prog.layers)):
        return False
    return True


def _new(dev, *shape):
    return torch.empty(shape, device=dev, dtype=torch.float32)


def _rows(t):
    """fp32 rows the kernels address directly (unit inner stride, 16-byte aligned rows); anything else is copied once."""
    if t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1 and t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0 and t.stride(0) >= t.shape[1]:
        return t
    return core.f32c(t)


def _seed(p):
    return core.next_seed() if p > 0 else 0


def _seed_ptr(p):
    return core.iter_seed().data_ptr() if p > 0 else None


def _launch(fn, struct, args, what, flops=0.0, nbytes=0.0):
    """One group launch; flops > 0: a contraction launch, accounted with the engine's (bench.py roofline leg, hip.core.engine_call)."""
    n = len(args)
    arr = (struct * n)(*args)
    if flops > 0:
        core.engine_call('ldetr_token_stack', flops, lambda: core.check(fn(arr, n, core.stream()), what), nbytes=nbytes)
    else:
        core.check(fn(arr, n, core.stream()), what)


def _fl_self(pr, bwd=False):
    """algorithmic FLOPs / bytes of the self-attention sub-block of one stack (projection, QK^T + PV, output projection; backward: their
    data gradients incl. the recomputed scores)."""
    M, D, att = pr.B * pr.L, D_MODEL, pr.B * N_HEAD * 2.0 * pr.L * pr.L * (D_MODEL // N_HEAD)
    fl = 2.0 * M * D * 3 * D + 2.0 * M * D * D + (5 if bwd else 2) * att
    return fl, 4.0 * (M * D * (6 if bwd else 5) + 4 * D * D + N_HEAD * M * D)


def _fl_cross(pr, bwd=False):
    M, D, att = pr.B * pr.L, D_MODEL, pr.B * N_HEAD * 2.0 * pr.L * pr.S * (D_MODEL // N_HEAD)
    fl = 2.0 * M * D * D * 2 + (5 if bwd else 2) * att
    return fl, 4.0 * (M * D * 4 + 2 * D * D + (4 if bwd else 2) * pr.B * pr.S * D + N_HEAD * M * D)


def _fl_ffn(M, F):
    return 4.0 * M * D_MODEL * F, 4.0 * (2 * D_MODEL * F + M * F + (F // 64 + 1) * M * D_MODEL)


# ---------------------------------------------------------------------------------------------------------------- argument blocks
def _ln_fwd_args(x, r, n_parts, r_bias, gamma, beta, eps, p_drop, seed, y, z, mean, rstd):
    a = _lib.LnArgs()
    M, D = x.shape
    a.x, a.gamma, a.beta, a.y, a.z, a.mean, a.rstd = x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), z.data_ptr(), mean.data_ptr(), rstd.data_ptr()
    a.r = r.data_ptr() if r is not None else None
    a.rows, a.D, a.eps = M, D, eps
    a.p_drop, a.seed, a.seed_ptr = (p_drop, seed, _seed_ptr(p_drop)) if r is not None else (0.0, 0, None)
    a.r_parts, a.r_part_stride = n_parts, M * D
    a.r_bias = r_bias.data_ptr() if (r_bias is not None and n_parts > 0) else None
    return a


def _ln_bwd_args(dy, dy2, parts, n_parts, z, mean, rstd, gamma, dx, dr, dgamma, dbeta, p_drop, seed):
    a = _lib.LnArgs()
    M, D = z.shape
    a.dy, a.z, a.mean, a.rstd, a.gamma, a.dx = dy.data_ptr(), z.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), dx.data_ptr()
    a.dy2 = dy2.data_ptr() if dy2 is not None else None
    a.dr = dr.data_ptr() if dr is not None else None
    a.dgamma = dgamma.data_ptr() if dgamma is not None else None
    a.dbeta = dbeta.data_ptr() if dbeta is not None else None
    a.dy_parts = parts.data_ptr() if n_parts else None
    a.dy_nparts, a.dy_part_stride = n_parts, M * D
    a.rows, a.D, a.p_drop, a.seed, a.seed_ptr = M, D, p_drop, seed, _seed_ptr(p_drop)
    return a


class _Grad(object):
    """A gradient inside the node: `res` [M, D] + the sum of `n` slices parts[s][M][D] (n = 0: res alone); dy2: one more full tensor (the
    generic cross-attention path's query-projection gradient)."""

    def __init__(self, res, parts=None, n=0, dy2=None):
        self.res, self.parts, self.n, self.dy2 = res, parts, n, dy2


class _TokenStacksFn(torch.autograd.Function):
    """forward(progs, *tensors) -> one output [B*L, 256] per stack.  tensors = per stack: x, every layer's parameters (layer_params order), the final
    norm's (weight, bias) if any, then for 'dec' K_0..K_{n-1}, V_0..V_{n-1} of hip.attention.grouped_kv."""

    @staticmethod
    def forward(ctx, progs, *tensors):
        ctx.set_materialize_grads(False)
        lib = core.lib()
        dev = progs[0].x.device
        core.require_gpu(*[t for t in tensors if t is not None])
        # ---- unpack
        st, pos = [], 0
        for pr in progs:
            s = dict(prog=pr, x_index=pos, M=pr.B * pr.L)
            s['cur'] = _rows(tensors[pos]); pos += 1
            npar = DEC_PARAMS if pr.kind == 'dec' else ENC_PARAMS
            s['lp'], s['lp_index'] = [], []
            for _ in pr.layers:
                s['lp'].append([core.f32c(t.detach()) for t in tensors[pos:pos + npar]]); s['lp_index'].append(pos); pos += npar
            if pr.final_norm is not None:
                s['fn'] = [core.f32c(t.detach()) for t in tensors[pos:pos + 2]]; s['fn_index'] = pos; pos += 2
            if pr.kind == 'dec':
                n = len(pr.layers)
                s['K'], s['V'], s['kv_index'] = list(tensors[pos:pos + n]), list(tensors[pos + n:pos + 2 * n]), pos
                pos += 2 * n
            s['saved'] = []
            st.append(s)
        assert pos == len(tensors)
        H, D = N_HEAD, D_MODEL
        scale = 1.0 / math.sqrt(D // H)
        nl = max(len(s['prog'].layers) for s in st)
        for i in range(nl):
            act = [s for s in st if i < len(s['prog'].layers)]
            # ---- self-attention sub-block of every active stack: ONE launch
            args = []
            for s in act:
                pr, M, W = s['prog'], s['M'], s['lp'][i]
                layer = pr.layers[i]
                p_att = layer.self_attn.dropout if pr.training else 0.0
                sv = dict(x=s['cur'], qkv=_new(dev, M, 3 * D), o=_new(dev, M, D), lse=_new(dev, pr.B * H * pr.L), p_att=p_att, seed_att=_seed(p_att))
                ypart = _new(dev, H, M, D)
                a = _lib.MhaSmallArgs()
                a.x, a.ldx, a.w_in, a.b_in, a.w_out = sv['x'].data_ptr(), sv['x'].stride(0), W[0].data_ptr(), W[1].data_ptr(), W[2].data_ptr()
                a.kpm = pr.kpm.data_ptr() if pr.kpm is not None else None
                a.qkv, a.o, a.lse, a.ypart, a.B, a.L = sv['qkv'].data_ptr(), sv['o'].data_ptr(), sv['lse'].data_ptr(), ypart.data_ptr(), pr.B, pr.L
                a.scale, a.p_drop, a.seed, a.seed_ptr = scale, p_att, sv['seed_att'], _seed_ptr(p_att)
                args.append(a)
                s['sv'], s['r'], s['r_bias'] = sv, ypart, W[3]
                s['saved'].append(sv)
            fb = [_fl_self(s['prog']) for s in act]
            _launch(lib.ldetr_mha_small_fwd_group_f32, _lib.MhaSmallArgs, args, 'mha_small_fwd', sum(f for f, _ in fb), sum(b for _, b in fb))
            # ---- decoders: norm1, then the cross-attention sub-block onto the projected memory
            for s in act:
                pr = s['prog']
                if pr.kind != 'dec':
                    continue
                layer, M, W, sv = pr.layers[i], s['M'], s['lp'][i], s['sv']
                p1 = layer.dropout1.p if pr.training else 0.0
                sv.update(p1=p1, seed1=_seed(p1), z1=_new(dev, M, D), mean1=_new(dev, M), rstd1=_new(dev, M), t1=_new(dev, M, D))
                _launch(lib.ldetr_layernorm_fwd_group_f32, _lib.LnArgs,
                        [_ln_fwd_args(s['cur'], s['r'], H, s['r_bias'], W[4], W[5], layer.norm1.eps, p1, sv['seed1'], sv['t1'], sv['z1'], sv['mean1'], sv['rstd1'])], 'layernorm_fwd')
                s['cur'] = sv['t1']
                K, V = s['K'][i], s['V'][i]
                ca = layer.multihead_attn
                p_c = ca.dropout if pr.training else 0.0
                sv.update(p_c=p_c, seed_c=_seed(p_c), qc=_new(dev, M, D), oc=_new(dev, M, D), lse_c=_new(dev, pr.B * H * pr.L))
                Wq, Bq = W[6][:D], W[7][:D]
                small = (1 <= pr.S <= MAX_CROSS_KEYS and K.stride(1) == 1 and V.stride(1) == 1 and K.stride(0) % 4 == 0 and V.stride(0) % 4 == 0
                         and K.data_ptr() % 16 == 0 and V.data_ptr() % 16 == 0)
                sv['cross_small'] = small
                if small:
                    ypart = _new(dev, H, M, D)
                    fc = _fl_cross(pr)
                    core.engine_call('ldetr_token_stack', fc[0], lambda: core.check(lib.ldetr_mha_cross_fwd_f32(
                        core.ptr(sv['t1']), D, core.ptr(Wq), core.ptr(Bq), core.ptr(K), K.stride(0), core.ptr(V), V.stride(0), core.ptr(W[8]), core.ptr(pr.mem_kpm),
                        core.ptr(sv['qc']), core.ptr(sv['oc']), core.ptr(sv['lse_c']), core.ptr(ypart), pr.B, pr.L, pr.S, D, H, scale, p_c, sv['seed_c'],
                        _seed_ptr(p_c), core.stream()), 'mha_cross_fwd'), nbytes=fc[1])
                    s['r'], s['r_bias'], s['r_parts'] = ypart, W[9], H
                else:
                    # more than 64 memory tokens (backgrounds above 256 x 256): projection, attention kernel, projection
                    core.gemm(sv['t1'], Wq, 0, 0, M, D, D, out=sv['qc'], ep=core.epilogue(col_bias=Bq))
                    core.check(lib.ldetr_attention_fwd_f32(
                        core.ptr(sv['qc']), D, core.ptr(K), K.stride(0), core.ptr(V), V.stride(0), core.ptr(pr.mem_kpm), core.ptr(sv['oc']), D, core.ptr(sv['lse_c']),
                        pr.B, H, pr.L, pr.S, D // H, scale, p_c, sv['seed_c'], _seed_ptr(p_c), 0, core.stream()), 'attention_fwd')
                    a2 = core.gemm(sv['oc'], W[8], 0, 0, M, D, D, ep=core.epilogue(col_bias=W[9]))
                    s['r'], s['r_bias'], s['r_parts'] = a2, None, 0
            # ---- tail of every active stack: norm_a(x + drop(r)), feed-forward, norm_b: three launches
            ln_a, ffn, ln_b = [], [], []
            for s in act:
                pr, M, W, sv = s['prog'], s['M'], s['lp'][i], s['sv']
                layer = pr.layers[i]
                dec = pr.kind == 'dec'
                o = 6 if dec else 0      # offset of (norm_a, linear1, linear2, norm_b) in the layer's parameter list: enc 4.., dec 10..
                ga, ba, w1, b1, w2, b2, gb, bb = W[4 + o], W[5 + o], W[6 + o], W[7 + o], W[8 + o], W[9 + o], W[10 + o], W[11 + o]
                norm_a, norm_b = (layer.norm2, layer.norm3) if dec else (layer.norm1, layer.norm2)
                drop_a, drop_b = (layer.dropout2, layer.dropout3) if dec else (layer.dropout1, layer.dropout2)
                p_a, p_h, p_b = (drop_a.p, layer.dropout.p, drop_b.p) if pr.training else (0.0, 0.0, 0.0)
                F = w1.shape[0]
                ns = F // 64
                sv.update(p_a=p_a, seed_a=_seed(p_a), p_h=p_h, seed_h=_seed(p_h), p_b=p_b, seed_b=_seed(p_b), F=F,
                          x1=_new(dev, M, D), za=_new(dev, M, D), mean_a=_new(dev, M), rstd_a=_new(dev, M), h=_new(dev, M, F),
                          zb=_new(dev, M, D), mean_b=_new(dev, M), rstd_b=_new(dev, M))
                parts, y = _new(dev, ns, M, D), _new(dev, M, D)
                r_parts = s.get('r_parts', H) if dec else H
                ln_a.append(_ln_fwd_args(s['cur'], s['r'], r_parts, s['r_bias'], ga, ba, norm_a.eps, p_a, sv['seed_a'], sv['x1'], sv['za'], sv['mean_a'], sv['rstd_a']))
                f = _lib.FfnArgs()
                f.x, f.ldx, f.w1, f.b1, f.w2, f.h, f.ypart = sv['x1'].data_ptr(), D, w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), sv['h'].data_ptr(), parts.data_ptr()
                f.M, f.F, f.p_drop, f.seed, f.seed_ptr = M, F, p_h, sv['seed_h'], _seed_ptr(p_h)
                ffn.append(f)
                ln_b.append(_ln_fwd_args(sv['x1'], parts, ns, b2, gb, bb, norm_b.eps, p_b, sv['seed_b'], y, sv['zb'], sv['mean_b'], sv['rstd_b']))
                s['keep'] = (s['r'], parts)      # (alive until their consumers have been queued: same stream, so queue order is enough)
                s['cur'] = y
            _launch(lib.ldetr_layernorm_fwd_group_f32, _lib.LnArgs, ln_a, 'layernorm_fwd')
            fb = [_fl_ffn(s['M'], s['sv']['F']) for s in act]
            _launch(lib.ldetr_ffn_fwd_group_f32, _lib.FfnArgs, ffn, 'ffn_fwd', sum(f for f, _ in fb), sum(b for _, b in fb))
            _launch(lib.ldetr_layernorm_fwd_group_f32, _lib.LnArgs, ln_b, 'layernorm_fwd')
        outs = []
        for s in st:
            pr = s['prog']
            if pr.final_norm is not None:
                M = s['M']
                fsv = dict(z=s['cur'], mean=_new(dev, M), rstd=_new(dev, M))
                y = _new(dev, M, D)
                _launch(lib.ldetr_layernorm_fwd_group_f32, _lib.LnArgs,
                        [_ln_fwd_args(s['cur'], None, 0, None, s['fn'][0], s['fn'][1], pr.final_norm.eps, 0.0, 0, y, s['cur'], fsv['mean'], fsv['rstd'])], 'layernorm_fwd')
                s['fsv'] = fsv
                s['cur'] = y
            outs.append(s['cur'])
        # ---- what the backward needs
        ctx.progs = progs
        ctx.params = tensors
        ctx.state = [dict(M=s['M'], x_index=s['x_index'], lp=s['lp'], lp_index=s['lp_index'], fn=s.get('fn'), fn_index=s.get('fn_index'),
                          K=s.get('K'), V=s.get('V'), kv_index=s.get('kv_index'), saved=s['saved'], fsv=s.get('fsv')) for s in st]
        return tuple(outs)

    @staticmethod
    def backward(ctx, *douts):
        lib = core.lib()
        progs, tensors, st = ctx.progs, ctx.params, ctx.state
        H, D = N_HEAD, D_MODEL
        grads = [None] * (1 + len(tensors))
        need = ctx.needs_input_grad
        wg_off = core.WEIGHT_GRADIENTS_DISABLED[0]
        temps = {}

        def target(idx):
            """Accumulation target of parameter tensors[idx]: its flat .grad view (the kernels add in place, autograd gets None) or a zero-filled
            temporary returned to autograd."""
            fg = core.flat_grad(tensors[idx])
            if fg is not None and fg.is_contiguous():
                return fg
            if idx not in temps:
                temps[idx] = torch.zeros_like(tensors[idx], memory_format=torch.contiguous_format)
                grads[1 + idx] = temps[idx]
            return temps[idx]

        live = []
        for s, pr, dy in zip(st, progs, douts):
            if dy is None:
                continue
            dev = dy.device
            npar = DEC_PARAMS if pr.kind == 'dec' else ENC_PARAMS
            s['need_w'] = (not wg_off) and any(need[1 + j] for base in s['lp_index'] for j in range(base, base + npar))
            g = _Grad(_rows(dy.reshape(s['M'], D)))
            if pr.final_norm is not None:
                fsv, M = s['fsv'], s['M']
                dx = _new(dev, M, D)
                want = (not wg_off) and (need[1 + s['fn_index']] or need[2 + s['fn_index']])
                _launch(lib.ldetr_layernorm_bwd_group_f32, _lib.LnArgs,
                        [_ln_bwd_args(g.res, None, None, 0, fsv['z'], fsv['mean'], fsv['rstd'], s['fn'][0], dx, None,
                                      target(s['fn_index']) if want else None, target(s['fn_index'] + 1) if want else None, 0.0, 0)], 'layernorm_bwd')
                g = _Grad(dx)
            s['g'] = g
            live.append((s, pr))
        if not live:
            return tuple(grads)
        dev = live[0][0]['g'].res.device
        nl = max(len(pr.layers) for _, pr in live)
        for i in reversed(range(nl)):
            act = [(s, pr) for s, pr in live if i < len(pr.layers)]
            # ---- tail backward: norm_b, feed-forward, norm_a
            a1, a2, a3 = [], [], []
            for s, pr in act:
                sv, M, W, base = s['saved'][i], s['M'], s['lp'][i], s['lp_index'][i]
                dec = pr.kind == 'dec'
                o = 6 if dec else 0
                ga, w1, w2, gb = W[4 + o], W[6 + o], W[8 + o], W[10 + o]
                nw = s['need_w']
                g = s['g']
                F, ns = sv['F'], sv['F'] // 64
                dz = _new(dev, M, D)
                dr = _new(dev, M, D) if sv['p_b'] > 0 else dz
                a1.append(_ln_bwd_args(g.res, g.dy2, g.parts, g.n, sv['zb'], sv['mean_b'], sv['rstd_b'], gb, dz, dr if sv['p_b'] > 0 else None,
                                       target(base + 10 + o) if nw else None, target(base + 11 + o) if nw else None, sv['p_b'], sv['seed_b']))
                dxpart, dh = _new(dev, ns, M, D), (_new(dev, M, F) if nw else None)
                f = _lib.FfnArgs()
                f.dy, f.x, f.ldx, f.h, f.w1, f.w2, f.dxpart = dr.data_ptr(), sv['x1'].data_ptr(), D, sv['h'].data_ptr(), w1.data_ptr(), w2.data_ptr(), dxpart.data_ptr()
                f.dh = dh.data_ptr() if dh is not None else None
                f.M, f.F, f.p_drop = M, F, sv['p_h']
                a2.append(f)
                dsum = _new(dev, M, D)
                da = _new(dev, M, D) if sv['p_a'] > 0 else dsum
                a3.append(_ln_bwd_args(dz, None, dxpart, ns, sv['za'], sv['mean_a'], sv['rstd_a'], ga, dsum, da if sv['p_a'] > 0 else None,
                                       target(base + 4 + o) if nw else None, target(base + 5 + o) if nw else None, sv['p_a'], sv['seed_a']))
                s['t'] = dict(dz=dz, dr=dr, dxpart=dxpart, dh=dh, dsum=dsum, da=da, gin=g)
            _launch(lib.ldetr_layernorm_bwd_group_f32, _lib.LnArgs, a1, 'layernorm_bwd')
            fb = [_fl_ffn(s['M'], s['saved'][i]['F']) for s, _ in act]
            _launch(lib.ldetr_ffn_bwd_group_f32, _lib.FfnArgs, a2, 'ffn_bwd', sum(f for f, _ in fb), sum(b for _, b in fb))
            _launch(lib.ldetr_layernorm_bwd_group_f32, _lib.LnArgs, a3, 'layernorm_bwd')
            # ---- decoders: cross-attention backward, then norm1's
            for s, pr in act:
                if pr.kind != 'dec':
                    continue
                sv, M, W, base, t = s['saved'][i], s['M'], s['lp'][i], s['lp_index'][i], s['t']
                nw = s['need_w']
                K, V = s['K'][i], s['V'][i]
                dk, dv = pr.kvs[i][2]()            # views into the grouped projection's gradient buffers (hip.attention.grouped_kv)
                scale = 1.0 / math.sqrt(D // H)
                Wq = W[6][:D]
                if sv['cross_small']:
                    dq = _new(dev, M, D)
                    dxpart_c = _new(dev, H, M, D)
                    a = _lib.MhaCrossArgs()
                    a.w_q, a.k, a.ldk, a.v, a.ldv, a.w_out = Wq.data_ptr(), K.data_ptr(), K.stride(0), V.data_ptr(), V.stride(0), W[8].data_ptr()
                    a.kpm = pr.mem_kpm.data_ptr() if pr.mem_kpm is not None else None
                    a.q, a.o, a.lse, a.B, a.Lq, a.Lk = sv['qc'].data_ptr(), sv['oc'].data_ptr(), sv['lse_c'].data_ptr(), pr.B, pr.L, pr.S
                    a.scale, a.p_drop, a.seed, a.seed_ptr = scale, sv['p_c'], sv['seed_c'], _seed_ptr(sv['p_c'])
                    a.dr, a.dq, a.dk, a.lddk, a.dv, a.lddv, a.dxpart = t['da'].data_ptr(), dq.data_ptr(), dk.data_ptr(), dk.stride(0), dv.data_ptr(), dv.stride(0), dxpart_c.data_ptr()
                    fc = _fl_cross(pr, bwd=True)
                    core.engine_call('ldetr_token_stack', fc[0], lambda: core.check(lib.ldetr_mha_cross_bwd_f32(_lib.ctypes.byref(a), core.stream()), 'mha_cross_bwd'), nbytes=fc[1])
                    g1 = _Grad(t['dsum'], dxpart_c, H)
                else:
                    d_o = core.gemm(t['da'], W[8], 0, 1, M, D, D)
                    dq = _new(dev, M, D)
                    core.check(lib.ldetr_attention_bwd_f32(
                        core.ptr(sv['qc']), D, core.ptr(K), K.stride(0), core.ptr(V), V.stride(0), core.ptr(pr.mem_kpm), core.ptr(sv['oc']), D, core.ptr(sv['lse_c']),
                        core.ptr(d_o), D, core.ptr(dq), D, core.ptr(dk), dk.stride(0), core.ptr(dv), dv.stride(0), pr.B, H, pr.L, pr.S, D // H, scale, sv['p_c'], sv['seed_c'],
                        _seed_ptr(sv['p_c']), 0, core.stream()), 'attention_bwd')
                    g1 = _Grad(t['dsum'], dy2=core.gemm(dq, Wq, 0, 1, M, D, D))
                t['dq'], t['dk'], t['dv'] = dq, dk, dv
                if need[1 + s['kv_index'] + i]:
                    grads[1 + s['kv_index'] + i] = dk
                if need[1 + s['kv_index'] + len(pr.layers) + i]:
                    grads[1 + s['kv_index'] + len(pr.layers) + i] = dv
                dsum1 = _new(dev, M, D)
                da1 = _new(dev, M, D) if sv['p1'] > 0 else dsum1
                _launch(lib.ldetr_layernorm_bwd_group_f32, _lib.LnArgs,
                        [_ln_bwd_args(g1.res, g1.dy2, g1.parts, g1.n, sv['z1'], sv['mean1'], sv['rstd1'], W[4], dsum1, da1 if sv['p1'] > 0 else None,
                                      target(base + 4) if nw else None, target(base + 5) if nw else None, sv['p1'], sv['seed1'])], 'layernorm_bwd')
                t['da_ca'], t['da'], t['dsum'] = t['da'], da1, dsum1
            # ---- self-attention backward of every active stack: ONE launch
            args = []
            for s, pr in act:
                sv, M, W, t = s['saved'][i], s['M'], s['lp'][i], s['t']
                t['dqkv'] = _new(dev, M, 3 * D) if s['need_w'] else None
                t['dxpart_sa'] = _new(dev, H, M, D)
                a = _lib.MhaSmallArgs()
                a.w_in, a.w_out, a.qkv, a.o, a.lse = W[0].data_ptr(), W[2].data_ptr(), sv['qkv'].data_ptr(), sv['o'].data_ptr(), sv['lse'].data_ptr()
                a.kpm = pr.kpm.data_ptr() if pr.kpm is not None else None
                a.B, a.L, a.scale, a.p_drop, a.seed, a.seed_ptr = pr.B, pr.L, 1.0 / math.sqrt(D // H), sv['p_att'], sv['seed_att'], _seed_ptr(sv['p_att'])
                a.dr, a.dxpart = t['da'].data_ptr(), t['dxpart_sa'].data_ptr()
                a.dqkv = t['dqkv'].data_ptr() if t['dqkv'] is not None else None
                args.append(a)
            fb = [_fl_self(pr, bwd=True) for _, pr in act]
            _launch(lib.ldetr_mha_small_bwd_group_f32, _lib.MhaSmallArgs, args, 'mha_small_bwd', sum(f for f, _ in fb), sum(b for _, b in fb))
            # ---- every weight gradient of the layer(s): contractions over the tokens, up to 8 per launch
            descs = []
            for s, pr in act:
                if not s['need_w']:
                    continue
                sv, M, W, base, t = s['saved'][i], s['M'], s['lp'][i], s['lp_index'][i], s['t']
                dec = pr.kind == 'dec'
                o = 6 if dec else 0

                def desc(A, Bm, widx, bidx, rows=None):
                    d = _lib.WgradDesc()
                    dW, db = target(widx), target(bidx)
                    nrows = A.shape[1] if rows is None else rows
                    d.A, d.lda, d.B, d.ldb, d.dW, d.ldw, d.db = A.data_ptr(), A.stride(0), Bm.data_ptr(), Bm.stride(0), dW.data_ptr(), dW.shape[1], db.data_ptr()
                    d.M, d.rows, d.cols = M, nrows, Bm.shape[1]
                    return d
                descs.append(desc(t['dr'], sv['h'], base + 8 + o, base + 9 + o))                 # dW2 += dr^T h, db2
                descs.append(desc(t['dh'], sv['x1'], base + 6 + o, base + 7 + o))                # dW1 += dh^T x1, db1
                descs.append(desc(t['da'], sv['o'], base + 2, base + 3))                         # self-attention out_proj
                descs.append(desc(t['dqkv'], sv['x'], base + 0, base + 1))                       # self-attention in_proj (packed q | k | v)
                if dec:
                    descs.append(desc(t['da_ca'], sv['oc'], base + 8, base + 9))                 # cross-attention out_proj
                    descs.append(desc(t['dq'], sv['t1'], base + 6, base + 7, rows=D))            # cross-attention query rows of in_proj (K / V rows: the grouped projection)
            for j in range(0, len(descs), 8):
                chunk = descs[j:j + 8]
                arr = (_lib.WgradDesc * len(chunk))(*chunk)
                wf = sum(2.0 * d.M * d.rows * d.cols for d in chunk)
                wb = sum(4.0 * (d.M * (d.rows + d.cols) + d.rows * d.cols) for d in chunk)
                core.engine_call('ldetr_token_stack', wf, lambda: core.check(lib.ldetr_wgrad_multi_f32(arr, len(chunk), core.stream()), 'wgrad_multi'), nbytes=wb)
            for s, pr in act:
                t = s['t']
                s['g'] = _Grad(t['dsum'], t['dxpart_sa'], H)
                s['t'] = None
        # ---- the stacks' input gradients leave the node as ONE tensor each
        for s, pr in live:
            if need[1 + s['x_index']]:
                g, M = s['g'], s['M']
                dx = _new(dev, M, D)
                core.check(lib.ldetr_sum_parts_f32(core.ptr(g.res), core.ptr(g.parts), g.n, M * D, core.ptr(dx), M * D, core.stream()), 'sum_parts')
                grads[1 + s['x_index']] = dx.reshape(tensors[s['x_index']].shape)
        return tuple(grads)


def run(progs):
    """Run one or two independent stacks (Prog) in lock-step -> list of outputs [B*L, 256]."""
    NODE_RUNS[0] += 1
    tensors = []
    for pr in progs:
        tensors.append(pr.x)
        for layer in pr.layers:
            tensors += layer_params(pr.kind, layer)
        if pr.final_norm is not None:
            tensors += [pr.final_norm.weight, pr.final_norm.bias]
        if pr.kind == 'dec':
            tensors += [kv[0] for kv in pr.kvs] + [kv[1] for kv in pr.kvs]
    return list(_TokenStacksFn.apply(progs, *tensors))
