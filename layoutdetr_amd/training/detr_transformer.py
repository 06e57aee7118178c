"""DETR encoder-decoder on the gfx950 kernels, interface-compatible with the reference
`training/detr_transformer.py` (Transformer :73-112, TransformerWithToken :22-70, encoder/decoder layers
:180-322; post-norm only, relu FFN).  Same constructor signatures, same parameter names (so UP-DETR
checkpoints and `state_dict`s load unchanged), same forward signature and return values.

Internals differ by design: activations are batch-first row-major [B*L, d] matrices (row = b*L + l),
every projection / FFN is one f32-MFMA GEMM with fused bias/relu/dropout, attention is one fused
kernel per call, and each `x = norm(x + dropout(sub(x)))` is one kernel.
"""
import copy
from typing import Optional

import torch
from torch import Tensor, nn

from ..hip import composite, core
from ..hip.attention import grouped_kv, mha_cross_kv, mha_forward
from ..hip import ffn as hffn
from ..hip import stacks as hstacks
from ..hip.layernorm import add_layernorm
from ..hip.linear import linear


class _MHAParams(nn.MultiheadAttention):
    """Parameter container with nn.MultiheadAttention's names/initialisation; forward is the HIP path."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError('use layoutdetr_amd.hip.attention.mha_forward')


def _mha(m: nn.MultiheadAttention, q2, k2, v2, B, Lq, Lk, kpm, training, same_qk=False, same_qkv=False, qk_pos=None, qk_in=None, kv_alias=False):
    """-> (attention block output, alias of q2 to feed the residual branch from[, aliases of k2 and v2])."""
    p = m.dropout if training else 0.0
    return mha_forward(q2, k2, v2, m.in_proj_weight, m.in_proj_bias, m.out_proj.weight, m.out_proj.bias, m.num_heads,
                       B, Lq, Lk, key_padding_mask=kpm, p_drop=p, same_qk=same_qk, same_qkv=same_qkv, qk_pos=qk_pos, passthru=True,
                       qk_in=qk_in, kv_alias=kv_alias)


_GROUP_KV = True

def _mask_u8(kpm):
    """bool key-padding mask -> the uint8 form the attention kernels read: a torch.bool tensor is one byte per element holding 0 / 1, so a
    contiguous mask is reinterpreted in place (no launch); only a strided one is copied."""
    if kpm is None or kpm.dtype == torch.uint8:
        return kpm
    if kpm.dtype == torch.bool:
        return (kpm if kpm.is_contiguous() else kpm.contiguous()).view(torch.uint8)
    return kpm.to(torch.uint8).contiguous()


def _ffn(layer, x2):
    """-> (FFN output, alias of x2 for the residual branch)."""
    p = layer.dropout.p if layer.training else 0.0
    if hffn.large_usable(x2, layer.linear1, layer.linear2):
        return hffn.ffn_large(x2, layer.linear1, layer.linear2, p)
    h, x2 = linear(x2, layer.linear1.weight, layer.linear1.bias, act=core.ACT_RELU, p_drop=p, passthru=True)
    return linear(h, layer.linear2.weight, layer.linear2.bias), x2


def _add_ln(norm: nn.LayerNorm, x2, r2, drop: nn.Dropout, training, pos=None, r_bias=None):
    return add_layernorm(x2, r2, norm.weight, norm.bias, norm.eps, drop.p if training else 0.0, pos=pos, r_bias=r_bias)


def _add_ln_ffn_add_ln(layer, norm_a: nn.LayerNorm, x2, r2, drop_a: nn.Dropout, norm_b: nn.LayerNorm, drop_b: nn.Dropout, pos=None, r_bias=None):
    """The tail of a layer: x1 = norm_a(x + dropout(r)); norm_b(x1 + dropout(linear2(dropout(relu(linear1(x1)))))) (detr_transformer.py:210-214
    / 280-285).  On the token counts of the decoder-side stacks ONE autograd node of 3 launches forward / 4 backward (hip/ffn.py: fused
    feed-forward launch, its partial sums reduced inside the LayerNorm launches); otherwise LayerNorm, two GEMMs, LayerNorm."""
    t = layer.training
    if hffn.usable(x2, layer.linear1, layer.linear2):
        return hffn.add_ln_ffn_add_ln(x2, r2, norm_a, drop_a.p if t else 0.0, layer.linear1, layer.linear2, norm_b, layer.dropout.p if t else 0.0,
                                      drop_b.p if t else 0.0, pos=pos, r_bias=r_bias)
    x2 = _add_ln(norm_a, x2, r2, drop_a, t, r_bias=r_bias)
    f, x2 = _ffn(layer, x2)
    return _add_ln(norm_b, x2, f, drop_b, t, pos=pos)


class TransformerEncoderLayer(nn.Module):
    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation='relu', normalize_before=False):
        super().__init__()
        if activation != 'relu' or normalize_before:
            raise NotImplementedError('only the post-norm relu configuration used by LayoutDETR is implemented')
        self.self_attn = _MHAParams(d_model, nhead, dropout=dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.dropout1 = nn.Dropout(dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.normalize_before = normalize_before

    def forward2d(self, x2, B, L, kpm, pos2, xpos2=None, emit_pos=False):
        """xpos2: x2 + pos2 when the producer already formed it (the previous layer's norm2 emits it: emit_pos) -> (y2[, y2 + pos2])."""
        if pos2 is None:      # (the <= 16-token stacks at d_model 256 never get here: TransformerEncoder.forward2d hands them to hip.stacks as a whole)
            a, x2 = _mha(self.self_attn, x2, x2, x2, B, L, L, kpm, self.training, same_qkv=True)
            return _add_ln_ffn_add_ln(self, self.norm1, x2, a, self.dropout1, self.norm2, self.dropout2)
        a, x2 = _mha(self.self_attn, x2, x2, x2, B, L, L, kpm, self.training, same_qk=True, qk_pos=pos2, qk_in=xpos2)
        return _add_ln_ffn_add_ln(self, self.norm1, x2, a, self.dropout1, self.norm2, self.dropout2, pos=pos2 if emit_pos else None)


class TransformerDecoderLayer(nn.Module):
    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation='relu', normalize_before=False):
        super().__init__()
        if activation != 'relu' or normalize_before:
            raise NotImplementedError('only the post-norm relu configuration used by LayoutDETR is implemented')
        self.self_attn = _MHAParams(d_model, nhead, dropout=dropout)
        self.multihead_attn = _MHAParams(d_model, nhead, dropout=dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.norm3 = nn.LayerNorm(d_model)
        self.dropout1 = nn.Dropout(dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.dropout3 = nn.Dropout(dropout)
        self.normalize_before = normalize_before

    def forward2d(self, t2, mem2, mem_pos2, B, Lq, S, tgt_kpm, mem_kpm, kv=None):
        """-> (t2, alias of mem2, alias of mem_pos2): the next layer reads the memory through the aliases (mha_forward kv_alias).
        kv = (K, V, grad_dst) from hip.attention.grouped_kv: the memory projections of this layer were made by the stack (one GEMM for
        all layers); mem2 / mem_pos2 are then not touched here."""
        a, t2 = _mha(self.self_attn, t2, t2, t2, B, Lq, Lq, tgt_kpm, self.training, same_qkv=True)
        t2 = _add_ln(self.norm1, t2, a, self.dropout1, self.training)
        if kv is not None:
            m = self.multihead_attn
            a, t2 = mha_cross_kv(t2, kv[0], kv[1], kv[2], m.in_proj_weight, m.in_proj_bias, m.out_proj.weight, m.out_proj.bias, m.num_heads, B, Lq, S,
                                 key_padding_mask=mem_kpm, p_drop=m.dropout if self.training else 0.0)
        else:
            a, t2, mem_pos2, mem2 = _mha(self.multihead_attn, t2, mem_pos2, mem2, B, Lq, S, mem_kpm, self.training, kv_alias=True)
        return _add_ln_ffn_add_ln(self, self.norm2, t2, a, self.dropout2, self.norm3, self.dropout3), mem2, mem_pos2


def _get_clones(module, N):
    return nn.ModuleList([copy.deepcopy(module) for _ in range(N)])


class TransformerEncoder(nn.Module):
    def __init__(self, encoder_layer, num_layers, norm=None):
        super().__init__()
        self.layers = _get_clones(encoder_layer, num_layers)
        self.num_layers = num_layers
        self.norm = norm

    def as_prog(self, x2, B, L, kpm):
        """This stack on x2 [B*L, d] as a hip.stacks.Prog (to run alone or in lock-step with an independent partner stack), or None when the stack
        node does not take it (position embeddings, a final norm, another width / more than 16 tokens per sample)."""
        if self.norm is not None or not hstacks.ENABLED:
            return None
        prog = hstacks.Prog('enc', self.layers, x2, B, L, _mask_u8(kpm), self.training)
        return prog if hstacks.usable(prog) else None

    def forward2d(self, x2, B, L, kpm, pos2, want_pos=False):
        """want_pos (with pos2): also return x_out + pos2 (the decoder's cross-attention key input, detr_transformer.py:277), formed by
        the last layer's LayerNorm launch.  Between layers the position-embedded copy comes from the previous layer's norm2 as well:
        one add launch per stack (the first layer's input) instead of one per layer."""
        kpm = _mask_u8(kpm)   # one conversion per stack, not one per attention launch
        if pos2 is None and not want_pos:
            prog = self.as_prog(x2, B, L, kpm)
            if prog is not None:
                return hstacks.run([prog])[0]
        xpos2 = None
        n = len(self.layers)
        for i, layer in enumerate(self.layers):
            emit = pos2 is not None and (i + 1 < n or (want_pos and self.norm is None))
            if pos2 is not None and xpos2 is None:
                xpos2 = x2 + pos2
            r = layer.forward2d(x2, B, L, kpm, pos2, xpos2=xpos2, emit_pos=emit)
            x2, xpos2 = r if emit else (r, None)
        if self.norm is not None:
            r = add_layernorm(x2, None, self.norm.weight, self.norm.bias, self.norm.eps, pos=pos2 if want_pos else None)
            x2, xpos2 = r if (want_pos and pos2 is not None) else (r, None)
        if want_pos:
            return x2, (xpos2 if xpos2 is not None else x2 + pos2)
        return x2


class TransformerDecoder(nn.Module):
    def __init__(self, decoder_layer, num_layers, norm=None, return_intermediate=False):
        super().__init__()
        if return_intermediate:
            raise NotImplementedError('return_intermediate_dec=True is not used by LayoutDETR')
        self.layers = _get_clones(decoder_layer, num_layers)
        self.num_layers = num_layers
        self.norm = norm
        self.return_intermediate = return_intermediate

    def forward2d(self, t2, mem2, pos2, B, Lq, S, tgt_kpm, mem_kpm, mem_pos2=None, partner=None):
        """partner: a hip.stacks.Prog of an INDEPENDENT stack to advance in lock-step with this one (D's unconditional encoder beside its layout
        decoder) -> (output, partner's output); run on its own when this stack does not take the stack node."""
        if mem_pos2 is None:
            mem_pos2 = mem2 + pos2
        if composite.active():      # regulariser phases (R1 / path length) differentiate this stack twice: hip/composite.py
            out = composite.decoder_forward2d(self, t2, mem2, mem_pos2, B, Lq, S, tgt_kpm, mem_kpm)
            return out if partner is None else (out, hstacks.run([partner])[0])
        tgt_kpm, mem_kpm = _mask_u8(tgt_kpm), _mask_u8(mem_kpm)
        # the memory is the same for every layer (detr_transformer.py:277-280 projects it per layer): all layers' K / V projections as two
        # GEMMs with N = layers * d, their backward as four (hip.attention._GroupedKVFn); _GROUP_KV = False restores the per-layer launches
        kvs = grouped_kv(mem_pos2, mem2, [l.multihead_attn for l in self.layers]) if (_GROUP_KV and len(self.layers) > 1) else None
        if kvs is not None and hstacks.ENABLED:
            prog = hstacks.Prog('dec', self.layers, t2, B, Lq, tgt_kpm, self.training, final_norm=self.norm, kvs=kvs, S=S, mem_kpm=mem_kpm)
            if hstacks.usable(prog):
                outs = hstacks.run([prog] + ([partner] if partner is not None else []))
                return outs[0] if partner is None else (outs[0], outs[1])
        # the stack node did not take this stack (> 16 queries, > 512 rows, another width): the per-layer launches on the K / V blocks already projected
        # above; a partner runs on its own
        for i, layer in enumerate(self.layers):
            t2, mem2, mem_pos2 = layer.forward2d(t2, mem2, mem_pos2, B, Lq, S, tgt_kpm, mem_kpm, kv=None if kvs is None else kvs[i])
        if self.norm is not None:
            t2 = add_layernorm(t2, None, self.norm.weight, self.norm.bias, self.norm.eps)
        return t2 if partner is None else (t2, hstacks.run([partner])[0])


def _rows_from_nchw(x):
    """[B, C, h, w] (any memory format) -> ([B*h*w, C] row-major, h*w).  Free for channels_last inputs."""
    B, C, h, w = x.shape
    return x.permute(0, 2, 3, 1).reshape(B * h * w, C), h * w


class Transformer(nn.Module):
    _with_token = False

    def __init__(self, d_model=512, nhead=8, num_encoder_layers=6, num_decoder_layers=6, dim_feedforward=2048,
                 dropout=0.1, activation='relu', normalize_before=False, return_intermediate_dec=False):
        super().__init__()
        if self._with_token:
            self.token = nn.Parameter(torch.randn(1, 1, d_model))
            self.register_buffer('token_mask', torch.zeros(1, 1, dtype=torch.bool))
        encoder_layer = TransformerEncoderLayer(d_model, nhead, dim_feedforward, dropout, activation, normalize_before)
        encoder_norm = nn.LayerNorm(d_model) if normalize_before else None
        self.encoder = TransformerEncoder(encoder_layer, num_encoder_layers, encoder_norm)
        decoder_layer = TransformerDecoderLayer(d_model, nhead, dim_feedforward, dropout, activation, normalize_before)
        decoder_norm = nn.LayerNorm(d_model)
        self.decoder = TransformerDecoder(decoder_layer, num_decoder_layers, decoder_norm,
                                          return_intermediate=return_intermediate_dec)
        self._reset_parameters()
        self.d_model = d_model
        self.nhead = nhead
        if d_model % nhead != 0 or (d_model // nhead) % 32 != 0 or d_model // nhead > 192:
            raise NotImplementedError('the fused attention kernels take head widths that are multiples of 32 up to 192 '
                                      f'(d_model {d_model} / {nhead} heads = {d_model / nhead:g})')

    def _reset_parameters(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def forward(self, src, mask, pos_embed, tgt, tgt_key_padding_mask, decoder_mask=None, partner=None):
        """partner (extension): a hip.stacks.Prog to run in lock-step with the decoder -> (hs, memory, partner's output)."""
        if decoder_mask is not None:
            raise NotImplementedError('decoder_mask (tgt_mask) is never passed on the LayoutDETR path')
        bs, c, h, w = src.shape
        x2, S = _rows_from_nchw(src)
        pos2, _ = _rows_from_nchw(pos_embed)
        pos2 = pos2.contiguous()
        mem_kpm = mask.flatten(1)
        mem2, mem_pos2 = self.encoder.forward2d(x2, bs, S, mem_kpm, pos2, want_pos=True)
        if self._with_token:
            tgt = torch.cat([self.token.expand(-1, bs, -1), tgt], dim=0)
            tgt_key_padding_mask = torch.cat([self.token_mask.expand(bs, -1), tgt_key_padding_mask], dim=1)
        Lq = tgt.shape[0]
        t2 = tgt.permute(1, 0, 2).reshape(bs * Lq, c)
        hs2 = self.decoder.forward2d(t2, mem2, pos2, bs, Lq, S, tgt_key_padding_mask, mem_kpm, mem_pos2=mem_pos2, partner=partner)
        if partner is not None:
            hs2, other = hs2
        hs = hs2.reshape(bs, Lq, c)
        memory = mem2.reshape(bs, h, w, c).permute(0, 3, 1, 2)
        return (hs, memory) if partner is None else (hs, memory, other)


class TransformerWithToken(Transformer):
    _with_token = True


def build_transformer(args):
    return Transformer(d_model=args.hidden_dim, dropout=args.dropout, nhead=args.nheads,
                       dim_feedforward=args.dim_feedforward, num_encoder_layers=args.enc_layers,
                       num_decoder_layers=args.dec_layers, normalize_before=args.pre_norm, return_intermediate_dec=True)
