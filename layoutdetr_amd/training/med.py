"""BERT text encoder of the LayoutDETR hot path, forward only, on the gfx950 kernels (first half).

Drop-in for the frozen `text_encoder` the reference builds at training/networks_detr.py:88-93 / :213-218
(`BertModel(config, add_pooling_layer=False)` from training/med.py) as it is CALLED on the hot path
(networks_detr.py:146,290):  `text_encoder(input_ids, attention_mask=..., return_dict=True, mode='text')`
followed by `.last_hidden_state[:, 0, :]`.  Same sub-module and parameter names as training/med.py
(BertEmbeddings :55-97, BertSelfAttention :100-228, BertSelfOutput :231-241, BertIntermediate :296-307, BertOutput
:310-320, BertLayer :323-386 in mode='text', BertEncoder :389-486), so a reference / HF `bert-base-uncased`
state_dict loads with strict=True (the cross-attention parameters of an `add_cross_attention` config exist and are never
touched in text mode, exactly as in the reference).

The module is frozen in the reference (`requires_grad_(False)`, training_loop.py:283): there is no backward path here,
and calling it with gradients enabled on its parameters raises.  Per layer: one packed q|k|v GEMM (the three weight
matrices are concatenated once and cached), fused attention for head widths 32..192 (`csrc/attention.hip`), output
projection, residual + dropout + LayerNorm in one kernel, bias + erf-GELU as one in-place streaming pass.
Tokenisation (strings -> ids) stays on the host and outside this package: pass token ids.
"""
import ctypes
import math
from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..hip import core
from ..hip.layernorm import add_layernorm


class BertConfig(SimpleNamespace):
    """The fields of configs/med_config.json that the text-mode forward reads (defaults = that file)."""

    def __init__(self, vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
                 max_position_embeddings=512, layer_norm_eps=1e-12, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1,
                 pad_token_id=0, add_cross_attention=True, encoder_width=768, **unused):
        super().__init__(vocab_size=vocab_size, hidden_size=hidden_size, num_hidden_layers=num_hidden_layers,
                         num_attention_heads=num_attention_heads, intermediate_size=intermediate_size,
                         max_position_embeddings=max_position_embeddings, layer_norm_eps=layer_norm_eps,
                         hidden_dropout_prob=hidden_dropout_prob, attention_probs_dropout_prob=attention_probs_dropout_prob,
                         pad_token_id=pad_token_id, add_cross_attention=add_cross_attention, encoder_width=encoder_width)

    @classmethod
    def from_json_file(cls, path):
        import json
        with open(path) as f:
            return cls(**json.load(f))


class _TokenEmbeddingFn(torch.autograd.Function):
    """word_embeddings(input_ids) + position_embeddings(position_ids[:, :T]) (training/med.py:88-94) as one gather kernel; the
    backward scatters with fp32 atomics straight into the (tied) vocabulary gradient instead of aten's sort-based
    embedding_dense_backward (whose rocPRIM sort produced wild indices under hipGraph replay once B*T exceeded 3072 tokens)."""

    @staticmethod
    def forward(ctx, input_ids, word, pos, padding_idx):
        core.require_gpu(input_ids, word, pos)
        B, T = input_ids.shape
        V, d = word.shape
        ids = input_ids.to(torch.int64).contiguous()
        wd, pd = word.detach().contiguous(), pos.detach().contiguous()
        out = torch.empty((B * T, d), device=word.device, dtype=torch.float32)
        core.check(core.lib().ldetr_embedding_fwd_f32(core.ptr(wd), core.ptr(pd), core.ptr(ids), core.ptr(out), B * T, d, V, T,
                                                      core.stream()), 'embedding_fwd')
        ctx.save_for_backward(ids)
        ctx.params = (word, pos)
        ctx.cfg = (B, T, V, d, -1 if padding_idx is None else int(padding_idx), pos.shape[0])
        return out

    @staticmethod
    def backward(ctx, dy):
        ids, = ctx.saved_tensors
        word, pos = ctx.params
        B, T, V, d, padding_idx, P = ctx.cfg
        dy = dy.to(torch.float32).contiguous()
        gw = gp = None
        if ctx.needs_input_grad[1]:
            acc = core.flat_grad(word)
            tgt = acc if acc is not None else torch.zeros((V, d), device=dy.device, dtype=torch.float32)
            core.check(core.lib().ldetr_embedding_bwd_f32(core.ptr(dy), core.ptr(ids), core.ptr(tgt), B * T, d, V, padding_idx,
                                                          core.stream()), 'embedding_bwd')
            gw = None if acc is not None else tgt
        if ctx.needs_input_grad[2]:
            gp = torch.zeros((P, d), device=dy.device, dtype=torch.float32)
            gp[:T] = dy.view(B, T, d).sum(0)
        return None, gw, gp, None


def token_embedding(emb, input_ids):
    """[B*T, hidden] rows of BertEmbeddings before its LayerNorm."""
    return _TokenEmbeddingFn.apply(input_ids, emb.word_embeddings.weight, emb.position_embeddings.weight, emb.word_embeddings.padding_idx)


class BertEmbeddings(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.word_embeddings = nn.Embedding(config.vocab_size, config.hidden_size, padding_idx=config.pad_token_id)
        self.position_embeddings = nn.Embedding(config.max_position_embeddings, config.hidden_size)
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)
        self.register_buffer('position_ids', torch.arange(config.max_position_embeddings).expand((1, -1)))

    def forward(self, input_ids):
        x = add_layernorm(token_embedding(self, input_ids), None, self.LayerNorm.weight, self.LayerNorm.bias, self.LayerNorm.eps)
        return self.dropout(x)   # [B*T, hidden]


class _SelfAttentionParams(nn.Module):
    def __init__(self, hidden, kv_width=None):
        super().__init__()
        self.query = nn.Linear(hidden, hidden)
        self.key = nn.Linear(kv_width or hidden, hidden)
        self.value = nn.Linear(kv_width or hidden, hidden)


class _SelfOutput(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)


class BertAttention(nn.Module):
    def __init__(self, config, is_cross_attention=False):
        super().__init__()
        self.self = _SelfAttentionParams(config.hidden_size, config.encoder_width if is_cross_attention else None)
        self.output = _SelfOutput(config)


class _Intermediate(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.intermediate_size)


class _Output(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.intermediate_size, config.hidden_size)
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)


class BertLayer(nn.Module):
    def __init__(self, config, layer_num):
        super().__init__()
        self.config = config
        self.layer_num = layer_num
        self.attention = BertAttention(config)
        if config.add_cross_attention:
            self.crossattention = BertAttention(config, is_cross_attention=True)   # mode='multimodal' only: unused here
        self.intermediate = _Intermediate(config)
        self.output = _Output(config)
        self._packed = None

    def _qkv(self):
        """q|k|v weights concatenated to one [3*hidden, hidden] matrix (+ bias), rebuilt only when a weight was modified."""
        a = self.attention.self
        ver = (a.query.weight._version, a.key.weight._version, a.value.weight._version, a.query.bias._version,
               a.key.bias._version, a.value.bias._version, a.query.weight.data_ptr())
        if self._packed is None or self._packed[0] != ver:
            w = torch.cat([a.query.weight, a.key.weight, a.value.weight], 0).detach().contiguous()
            b = torch.cat([a.query.bias, a.key.bias, a.value.bias], 0).detach().contiguous()
            self._packed = (ver, w, b)
        return self._packed[1], self._packed[2]

    def forward2d(self, x2, B, T, kpm):
        cfg = self.config
        d, H = cfg.hidden_size, cfg.num_attention_heads
        dh = d // H
        M = x2.shape[0]
        p_attn = cfg.attention_probs_dropout_prob if self.training else 0.0
        p_hid = cfg.hidden_dropout_prob if self.training else 0.0
        w, b = self._qkv()
        qkv = core.gemm(x2, w, 0, 0, M, 3 * d, d, ep=core.epilogue(col_bias=b))
        ctx = torch.empty((M, d), device=x2.device, dtype=torch.float32)
        seed = core.next_seed() if p_attn > 0 else 0
        q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
        core.check(core.lib().ldetr_attention_fwd_f32(
            core.ptr(q), qkv.stride(0), core.ptr(k), qkv.stride(0), core.ptr(v), qkv.stride(0), core.ptr(kpm),
            core.ptr(ctx), d, None, B, H, T, T, dh, 1.0 / math.sqrt(dh), p_attn, seed,
            core.seed_ptr() if p_attn > 0 else None, 0, core.stream()), 'bert attention')
        so = self.attention.output
        a = core.gemm(ctx, so.dense.weight.detach(), 0, 0, M, d, d, ep=core.epilogue(col_bias=so.dense.bias.detach()))
        x2 = add_layernorm(x2, a, so.LayerNorm.weight, so.LayerNorm.bias, so.LayerNorm.eps, p_hid)
        it, out = self.intermediate.dense, self.output
        h = core.gemm(x2, it.weight.detach(), 0, 0, M, cfg.intermediate_size, d)
        # bias + erf-GELU in place, one streaming pass (inside the contraction kernel's epilogue the erf cost 320 bytes of
        # scratch per lane and sent the accumulator tile through it: 14x slower GEMMs)
        core.check(core.lib().ldetr_bias_act_f32(core.ptr(h), core.ptr(it.bias.detach()), None, None, None, core.ptr(h), h.numel(),
                                                 cfg.intermediate_size, 1, 0, 10, 0.0, 1.0, -1.0, core.stream()), 'bias_act gelu')
        f = core.gemm(h, out.dense.weight.detach(), 0, 0, M, d, cfg.intermediate_size, ep=core.epilogue(col_bias=out.dense.bias.detach()))
        return add_layernorm(x2, f, out.LayerNorm.weight, out.LayerNorm.bias, out.LayerNorm.eps, p_hid)


class BertEncoder(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.layer = nn.ModuleList([BertLayer(config, i) for i in range(config.num_hidden_layers)])


class BertModel(nn.Module):
    """Text-mode forward of the reference's BertModel(add_pooling_layer=False)."""

    def __init__(self, config, add_pooling_layer=False):
        super().__init__()
        if add_pooling_layer:
            raise NotImplementedError('the hot path builds the text encoder with add_pooling_layer=False (networks_detr.py:92)')
        self.config = config
        self.embeddings = BertEmbeddings(config)
        self.encoder = BertEncoder(config)

    def forward(self, input_ids, attention_mask=None, return_dict=True, mode='text'):
        if mode != 'text':
            raise NotImplementedError("only mode='text' (self-attention only) is on the hot path (training/med.py:361)")
        core.require_gpu(input_ids)
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise RuntimeError('the text encoder is forward-only (frozen in the reference, training_loop.py:283): '
                               'call text_encoder.requires_grad_(False) or run it under torch.no_grad()')
        B, T = input_ids.shape
        if T > 256:
            raise NotImplementedError('sequence length > 256 is not supported by the fused attention kernel')
        with torch.no_grad():
            # reference: additive mask (1 - attention_mask) * -10000 (get_extended_attention_mask); a masked key's probability
            # underflows to exactly 0 in fp32 either way, so the kernel's -inf key-padding mask gives the same numbers
            kpm = None if attention_mask is None else (attention_mask == 0).to(torch.uint8).contiguous()
            x2 = self.embeddings(input_ids)
            for layer in self.encoder.layer:
                x2 = layer.forward2d(x2, B, T, kpm)
            hs = x2.reshape(B, T, -1)
        return SimpleNamespace(last_hidden_state=hs) if return_dict else (hs,)


# ----------------------------------------------------------------------------------------------------------------------
# LM text decoder (second half): the reference's `BertLMHeadModel` as the hot path calls it
# (networks_detr.py:169-181, 328-340): `text_decoder(decoder_input_ids, attention_mask=..., encoder_hidden_states=xx,
# labels=decoder_targets, return_dict=True, mode='text')`.  With mode='text' BertLayer skips the cross-attention block
# (training/med.py:361), so `encoder_hidden_states` is accepted and ignored exactly as in the reference: the decoder is a causal
# BERT LM over the text tokens (is_decoder=True -> causal + padding mask, med.py:704-739), trained with a label-smoothed (0.1)
# next-token cross entropy (med.py:911-916).  Trainable: every op below has a backward on the HIP kernels (GEMM engine, causal
# wide-head attention fwd/bwd, fused add+dropout+LayerNorm, erf-GELU fwd/grad, label-smoothed softmax cross entropy over the
# vocabulary, csrc/xent.hip); only the embedding lookups are torch glue.
class _GeluFn(torch.autograd.Function):
    """y = gelu(h + bias) with the erf form; backward through bias_act's gradient kernel (activation 10)."""

    @staticmethod
    def forward(ctx, h, bias):
        core.require_gpu(h, bias)
        h = core.f32c(h)
        y = torch.empty_like(h)
        C = h.shape[-1]
        core.check(core.lib().ldetr_bias_act_f32(core.ptr(h), core.ptr(bias), None, None, None, core.ptr(y), h.numel(), C, 1, 0, 10,
                                                 0.0, 1.0, -1.0, core.stream()), 'gelu fwd')
        ctx.save_for_backward(h, bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        h, bias = ctx.saved_tensors
        dy = core.f32c(dy)
        dh = torch.empty_like(h)
        C = h.shape[-1]
        core.check(core.lib().ldetr_bias_act_f32(core.ptr(dy), core.ptr(bias), core.ptr(h), None, None, core.ptr(dh), h.numel(), C, 1, 1, 10,
                                                 0.0, 1.0, -1.0, core.stream()), 'gelu bwd')
        db = core.colsum(dh.reshape(-1, C)).reshape(-1) if ctx.needs_input_grad[1] else None
        return dh, db


class _SoftmaxXentFn(torch.autograd.Function):
    """mean_i over targets != ignore_index of the label-smoothed cross entropy; logits [rows, V] fp32, targets [rows] int64.
    The gradient overwrites nothing the forward needs except the logits themselves, which autograd does not reuse after this
    node: it is written into a fresh buffer (logits may be another node's saved output)."""

    @staticmethod
    def forward(ctx, logits, targets, ignore_index, label_smoothing, raw=False):
        """raw: return the (loss sum, count) pair instead of their ratio (hip.losses.combine forms the ratio with the other terms of a phase)."""
        core.require_gpu(logits, targets)
        if logits.dtype != torch.float32 or logits.stride(1) != 1 or logits.stride(0) % 4 != 0 or logits.data_ptr() % 16 != 0:
            # rows must start 16-byte aligned: copy into a padded pitch (never taken for the 30524-entry vocabulary)
            buf = torch.empty((logits.shape[0], (logits.shape[1] + 3) // 4 * 4), device=logits.device, dtype=torch.float32)
            buf[:, :logits.shape[1]] = logits
            logits = buf[:, :logits.shape[1]]
        targets = targets.to(torch.int64).contiguous()
        rows, V = logits.shape
        lse = torch.empty(rows, device=logits.device, dtype=torch.float32)
        acc = core.zeros((2,), logits.device)      # [loss_sum, count]
        core.check(core.lib().ldetr_softmax_xent_fwd_f32(core.ptr(logits), logits.stride(0), core.ptr(targets), core.ptr(lse),
                                                         ctypes.c_void_p(acc.data_ptr()), ctypes.c_void_p(acc.data_ptr() + 4), rows, V,
                                                         ignore_index, label_smoothing, core.stream()), 'softmax_xent_fwd')
        ctx.save_for_backward(logits, targets, lse, acc)
        ctx.cfg = (ignore_index, label_smoothing, raw)
        return acc if raw else acc[0] / acc[1]

    @staticmethod
    def backward(ctx, g):
        logits, targets, lse, acc = ctx.saved_tensors
        ignore_index, label_smoothing, raw = ctx.cfg
        rows, V = logits.shape
        g = g.to(torch.float32).contiguous()        # (raw: g[0] is the gradient of the loss sum / count ratio's numerator side, see hip.losses)
        dx = torch.empty((rows, logits.stride(0)), device=logits.device, dtype=torch.float32)[:, :V]   # same (padded) row pitch
        core.check(core.lib().ldetr_softmax_xent_bwd_f32(core.ptr(logits), logits.stride(0), core.ptr(targets), core.ptr(lse),
                                                         ctypes.c_void_p(acc.data_ptr() + 4), core.ptr(g), core.ptr(dx), dx.stride(0), rows, V,
                                                         ignore_index, label_smoothing, core.stream()), 'softmax_xent_bwd')
        return dx, None, None, None, None


def softmax_cross_entropy(logits, targets, ignore_index=-100, label_smoothing=0.0, raw=False):
    return _SoftmaxXentFn.apply(logits, targets, ignore_index, label_smoothing, raw)


def _layer_train(layer, x2, B, T, kpm, causal):
    """Differentiable twin of BertLayer.forward2d (autograd Functions of hip/linear.py, hip/attention.py, hip/layernorm.py)."""
    from ..hip.attention import _AttnPackedFn
    from ..hip.linear import linear
    cfg = layer.config
    H = cfg.num_attention_heads
    p_attn = cfg.attention_probs_dropout_prob if layer.training else 0.0
    p_hid = cfg.hidden_dropout_prob if layer.training else 0.0
    a = layer.attention.self
    qkv = linear(x2, torch.cat([a.query.weight, a.key.weight, a.value.weight], 0), torch.cat([a.query.bias, a.key.bias, a.value.bias], 0))
    ctx = _AttnPackedFn.apply(qkv, None, kpm, B, H, T, p_attn, causal)
    so = layer.attention.output
    x2 = add_layernorm(x2, linear(ctx, so.dense.weight, so.dense.bias), so.LayerNorm.weight, so.LayerNorm.bias, so.LayerNorm.eps, p_hid)
    h = _GeluFn.apply(linear(x2, layer.intermediate.dense.weight, None), layer.intermediate.dense.bias)
    out = layer.output
    return add_layernorm(x2, linear(h, out.dense.weight, out.dense.bias), out.LayerNorm.weight, out.LayerNorm.bias, out.LayerNorm.eps, p_hid)


class _PredictionHeadTransform(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)


class _LMPredictionHead(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.transform = _PredictionHeadTransform(config)
        self.decoder = nn.Linear(config.hidden_size, config.vocab_size, bias=False)
        self.bias = nn.Parameter(torch.zeros(config.vocab_size))


class BertOnlyMLMHead(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.predictions = _LMPredictionHead(config)


class BertLMHeadModel(nn.Module):
    """Parameter names of the reference's BertLMHeadModel: `bert.*`, `cls.predictions.{transform.dense, transform.LayerNorm,
    decoder, bias}`; `cls.predictions.decoder.weight` is tied to `bert.embeddings.word_embeddings.weight` (HF tie_weights)."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.bert = BertModel(config, add_pooling_layer=False)
        self.cls = BertOnlyMLMHead(config)
        self.cls.predictions.decoder.weight = self.bert.embeddings.word_embeddings.weight

    def forward(self, input_ids, attention_mask=None, encoder_hidden_states=None, encoder_attention_mask=None, labels=None,
                return_dict=True, mode='text', reduction='mean'):
        from ..hip.linear import linear
        if mode != 'text':
            raise NotImplementedError("the hot path calls the text decoder with mode='text' (cross-attention skipped, med.py:361)")
        core.require_gpu(input_ids)
        cfg = self.config
        B, T = input_ids.shape
        emb = self.bert.embeddings
        x2 = add_layernorm(token_embedding(emb, input_ids), None, emb.LayerNorm.weight, emb.LayerNorm.bias, emb.LayerNorm.eps)
        x2 = emb.dropout(x2)
        kpm = None if attention_mask is None else (attention_mask == 0).to(torch.uint8).contiguous()
        for layer in self.bert.encoder.layer:
            x2 = _layer_train(layer, x2, B, T, kpm, True)
        hs = x2.reshape(B, T, cfg.hidden_size)
        # next-token prediction: scores of positions 0..T-2 against tokens 1..T-1 (med.py:911-913); only those rows reach the
        # vocabulary-sized GEMM
        hsh = hs[:, :-1].reshape(-1, cfg.hidden_size)
        tr = self.cls.predictions.transform
        t = _GeluFn.apply(linear(hsh, tr.dense.weight, None), tr.dense.bias)
        t = add_layernorm(t, None, tr.LayerNorm.weight, tr.LayerNorm.bias, tr.LayerNorm.eps)
        logits = linear(t, self.cls.predictions.decoder.weight, self.cls.predictions.bias)        # [B*(T-1), vocab]
        loss = None
        if labels is not None:
            tgt = labels[:, 1:].reshape(-1)
            if reduction != 'mean':
                raise NotImplementedError("the hot path uses reduction='mean' (training/med.py:886 default)")
            loss = softmax_cross_entropy(logits, tgt, ignore_index=-100, label_smoothing=0.1)
        return SimpleNamespace(loss=loss, logits=logits.reshape(B, T - 1, -1)) if return_dict else (loss, logits)
