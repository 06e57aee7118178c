"""LayoutDETR Generator / Discriminator on the gfx950 hot path — drop-in for the reference
`training/networks_detr.py` (Generator :65-187, Discriminator :190-361): same class names, constructor
keyword arguments, forward signatures, return tuples, parameter names and `text_encoder` attribute, so
`dnnlib.util.construct_class_by_name(class_name='layoutdetr_amd.training.networks_detr.Generator', ...)`
works from the reference `train.py` / `training_loop.py` unchanged (SURVEY §8b seam 1).

Text path (SURVEY §8a rows a16/a17 are *boundary inputs*, not kernel rows): `bbox_text` may be
  * a `TextFeatures` carrying the frozen BERT CLS features [B, N, 768] and character counts [B, N]
    (hot-path-only mode, BASELINE.md variant A), or
  * a `TextTokens` carrying the tokenizer's output (`input_ids`, `attention_mask` [B, N, T]) and the character counts —
    requires `text_mode='encoder'`: the frozen BERT text encoder (training/med.py) then runs on the HIP
    kernels inside forward(), exactly where the reference calls it (networks_detr.py:145-147, 289-291).
  * the reference's own form, a list (batch) of lists (elements) of strings (networks_detr.py:145), when the module was built with
    `tokenizer_vocab=<path to bert-base-uncased vocab.txt>` (or LDETR_BERT_VOCAB is set): training/tokenizer.py (host-side WordPiece,
    id-for-id equal to transformers.BertTokenizer, no download) turns them into `TextTokens`; needs text_mode 'encoder' / 'encoder+lm'.  `text_mode='encoder+lm'` additionally builds the trainable LM text decoder
(`text_decoder`, training/med.py BertLMHeadModel) and returns its label-smoothed next-token loss as `loss_lm`; in the other
modes `loss_lm` is a zero tensor.

`module.static_shapes = True` (opt-in, used by bench.py) switches the reconstruction heads from the reference's
boolean gathers `x[~padding_mask]` (dynamic shape M -> device-to-host sync, SURVEY §7 "launch-bound regime") to
full-slot tensors `[B, N, ...]` whose padded slots are masked out inside the losses: identical values up to summation
order, no host synchronisation, so a whole phase can be captured into a hipGraph.  The default (False) keeps the
reference's return shapes exactly.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..detr_util.misc import NestedTensor, nested_tensor_from_tensor_list
from ..hip import composite
from ..hip import conv as hconv
from ..hip import core
from ..hip.linear import linear
from .detr_backbone import Backbone, Joiner
from .detr_position_encoding import PositionEmbeddingSine
from .detr_transformer import Transformer, TransformerEncoder, TransformerEncoderLayer, TransformerWithToken
from .networks_stylegan2 import Decoder
from .util import TransformerWithToken_layoutganpp, encode_seq_first, encode_seq_first_pair


def merge_lists(lists):
    ret = []
    for l in lists:
        ret += l
    return ret


def split_list(list_a, chunk_size):
    return [list_a[i:i + chunk_size] for i in range(0, len(list_a), chunk_size)]


def normalize_2nd_moment(x, eps=1e-8):
    return x * (x.square().mean(dim=1, keepdim=True) + eps).rsqrt()


# Set by layoutdetr_amd.dropin.install(): the modules are being driven by the reference's own train.py / training_loop, which builds G / D
# from (class_name, z_dim, f_dim, ..., bert_* , im_f_dim) only (train.py:201-203,250-261) and hands them strings.  The reference ALWAYS
# builds the BERT tokenizer, the frozen text encoder and the LM text decoder (networks_detr.py:84-131), so under that driver an
# unspecified `text_mode` means 'encoder+lm' with the vocabulary taken from LDETR_BERT_VOCAB; stand-alone the default stays 'features'
# (text features are the hot path's boundary input).
REFERENCE_DEFAULTS = False


def _resolve_text_mode(text_mode, tokenizer):
    if text_mode is not None:
        return text_mode
    if not REFERENCE_DEFAULTS:
        return 'features'
    if tokenizer is None:
        raise RuntimeError("Generator / Discriminator built through the reference's module names (layoutdetr_amd.dropin) take strings as bbox_text "
                           "and therefore need the BERT vocabulary: set LDETR_BERT_VOCAB=<path to bert-base-uncased vocab.txt> (or pass "
                           "tokenizer_vocab=...), or pass text_mode='features' and feed TextFeatures")
    return 'encoder+lm'


def build_backbone():
    backbone = Backbone(name='resnet50', train_backbone=True, return_interm_layers=None, dilation=False)
    position_embedding = PositionEmbeddingSine(num_pos_feats=128, normalize=True)
    model = Joiner(backbone, position_embedding)
    model.num_channels = backbone.num_channels
    return model


class TextFeatures(object):
    """Precomputed output of the frozen text encoder for one batch (boundary input of the hot path)."""

    def __init__(self, text_feat, text_len, input_ids=None, attention_mask=None):
        self.text_feat = text_feat      # [B, N, bert_f_dim] fp32
        self.text_len = text_len        # [B, N] int64 (character counts, < max_text_length)
        self.input_ids = input_ids
        self.attention_mask = attention_mask

    def __len__(self):
        return self.text_feat.shape[0]

    def __getitem__(self, idx):  # supports the `bbox_text[:batch_size]` slicing of loss.py:125
        return TextFeatures(self.text_feat[idx], self.text_len[idx],
                            None if self.input_ids is None else self.input_ids[idx],
                            None if self.attention_mask is None else self.attention_mask[idx])


class TextTokens(object):
    """Tokenizer output for one batch: input_ids / attention_mask [B, N, T] int64, text_len [B, N] (character counts)."""

    def __init__(self, input_ids, attention_mask, text_len, bos_token_id=30522, pad_token_id=0):
        self.input_ids, self.attention_mask, self.text_len = input_ids, attention_mask, text_len
        # blip.init_tokenizer (:190-195) appends [DEC] (bos) and [ENC] to bert-base-uncased: ids 30522 / 30523, vocabulary 30524
        self.bos_token_id, self.pad_token_id = bos_token_id, pad_token_id

    def __len__(self):
        return self.input_ids.shape[0]

    def __getitem__(self, idx):
        return TextTokens(self.input_ids[idx], self.attention_mask[idx], self.text_len[idx], self.bos_token_id, self.pad_token_id)


def _build_text_encoder(text_mode, med_config, num_layers, num_heads):
    """reference: networks_detr.py:88-93 (encoder_config from med_config.json with layers / heads overridden)."""
    if text_mode == 'features':
        return _NoTextEncoder()
    if text_mode not in ('encoder', 'encoder+lm'):
        raise ValueError("text_mode must be 'features' (TextFeatures in), 'encoder' or 'encoder+lm' (TextTokens in)")
    import os
    from . import med
    cfg = med.BertConfig.from_json_file(med_config) if (med_config and os.path.exists(med_config)) else med.BertConfig()
    cfg.num_hidden_layers, cfg.num_attention_heads = num_layers, num_heads
    return med.BertModel(cfg, add_pooling_layer=False)


def _build_text_decoder(text_mode, med_config, num_layers, num_heads, encoder_width, vocab_size=30524):
    """reference: networks_detr.py:122-128 (decoder_config; BertLMHeadModel resized to the tokenizer's 30524 entries)."""
    if text_mode != 'encoder+lm':
        return None
    import os
    from . import med
    cfg = med.BertConfig.from_json_file(med_config) if (med_config and os.path.exists(med_config)) else med.BertConfig()
    cfg.num_hidden_layers, cfg.num_attention_heads, cfg.encoder_width, cfg.vocab_size = num_layers, num_heads, encoder_width, vocab_size
    return med.BertLMHeadModel(cfg)


def _lm_loss(module, bbox_text, padding_mask, B, N, static):
    """Text reconstruction loss of the reconstructor heads (reference networks_detr.py:169-181 / 328-340): the text decoder in
    mode='text' on the element texts with [DEC] as first token; padded slots contribute nothing (the reference gathers
    `[~padding_mask]` rows; here their labels are -100, the same mean over the same tokens, without a dynamic shape)."""
    if module.text_decoder is None or not isinstance(bbox_text, TextTokens):
        return None
    dev = padding_mask.device
    T = bbox_text.input_ids.shape[-1]
    ids = bbox_text.input_ids.reshape(B * N, T).to(dev).clone()
    am = bbox_text.attention_mask.reshape(B * N, T).to(dev)
    ids[:, 0] = bbox_text.bos_token_id
    targets = ids.masked_fill(ids == bbox_text.pad_token_id, -100)
    pm = padding_mask.reshape(B * N)
    if static:
        targets = targets.masked_fill(pm[:, None], -100)
    else:
        keep = ~pm
        ids, am, targets = ids[keep], am[keep], targets[keep]
    return module.text_decoder(ids, attention_mask=am, labels=targets, return_dict=True, mode='text').loss


class _NoTextEncoder(nn.Module):
    """Placeholder so `module.text_encoder.requires_grad_(False)` (training_loop.py:283) keeps working."""

    def forward(self, *a, **k):
        raise NotImplementedError("text_mode='features': pass TextFeatures instead of strings (BERT text path is SURVEY §8f-1)")


class _EmbeddingFn(torch.autograd.Function):
    """nn.Embedding lookup (emb_label / enc_text_len, networks_detr.py:85, 94) with the gradient scattered straight into the table's flat .grad view
    (fp32 atomics, csrc/xent.hip): aten's embedding_dense_backward zero-fills a table-sized temporary, scatters into it and AccumulateGrad adds it."""

    @staticmethod
    def forward(ctx, ids, weight):
        core.require_gpu(ids, weight)
        V, d = weight.shape
        i64 = ids.to(torch.int64).contiguous()
        out = torch.empty(tuple(ids.shape) + (d,), device=weight.device, dtype=torch.float32)
        core.check(core.lib().ldetr_embedding_fwd_f32(core.ptr(weight.detach().contiguous()), None, core.ptr(i64), core.ptr(out), i64.numel(), d, V, 1, core.stream()), 'embedding_fwd')
        ctx.save_for_backward(i64)
        ctx.param = weight
        return out

    @staticmethod
    def backward(ctx, dy):
        i64, = ctx.saved_tensors
        w = ctx.param
        V, d = w.shape
        acc = core.flat_grad(w)
        tgt = acc if (acc is not None and acc.is_contiguous()) else torch.zeros((V, d), device=dy.device, dtype=torch.float32)
        core.check(core.lib().ldetr_embedding_bwd_f32(core.ptr(core.f32c(dy)), core.ptr(i64), core.ptr(tgt), i64.numel(), d, V, -1, core.stream()), 'embedding_bwd')
        return None, (None if tgt is acc else tgt)


class Embedding(nn.Embedding):
    """nn.Embedding parameters (same names / initialisation); GPU lookups and their gradient on the HIP gather / scatter kernels."""

    def forward(self, ids):
        if ids.is_cuda and self.weight.dtype == torch.float32 and self.weight.shape[1] % 4 == 0 and self.padding_idx is None:
            return _EmbeddingFn.apply(ids, self.weight)
        return super().forward(ids)


class Linear(nn.Linear):
    """nn.Linear parameters, f32-MFMA forward/backward."""

    def forward(self, x, relu=False):
        if composite.active():      # regulariser phases (R1 / path length): a twice-differentiable node (hip/composite.py)
            return composite.linear(x, self.weight, self.bias, relu=relu)
        return linear(x, self.weight, self.bias, act=core.ACT_RELU if relu else core.ACT_NONE)


class Conv1x1(nn.Conv2d):
    """nn.Conv2d(cin, cout, kernel_size=1) parameters; NHWC implicit-GEMM forward/backward."""

    def forward(self, x_nchw_view):
        x = x_nchw_view.permute(0, 2, 3, 1)
        y = hconv.conv2d_nhwc(x, self.weight, None, self.bias, None, 1, 0, relu=False)
        return y.permute(0, 3, 1, 2)


class MLP(nn.Module):
    """Very simple multi-layer perceptron (reference: networks_detr.py:50-62); ReLU fused in the GEMM epilogue."""

    def __init__(self, input_dim, hidden_dim, output_dim, num_layers):
        super().__init__()
        self.num_layers = num_layers
        h = [hidden_dim] * (num_layers - 1)
        self.layers = nn.ModuleList(Linear(n, k) for n, k in zip([input_dim] + h, h + [output_dim]))

    def forward(self, x, final_relu=False):
        for i, layer in enumerate(self.layers):
            x = layer(x, relu=(i < self.num_layers - 1) or final_relu)
        return x


def _build_tokenizer(tokenizer_vocab):
    import os
    path = tokenizer_vocab or os.environ.get('LDETR_BERT_VOCAB')
    if not path:
        return None
    from .tokenizer import BertWordPieceTokenizer
    return BertWordPieceTokenizer(path)


def _coerce_text(module, bbox_text, device):
    """strings (the reference's input form) -> TextTokens through the module's host tokenizer; other forms pass through."""
    if isinstance(bbox_text, (list, tuple)) and len(bbox_text) and isinstance(bbox_text[0], (list, tuple)):
        if module.tokenizer is None:
            raise NotImplementedError('bbox_text strings need the BERT vocabulary: construct the module with tokenizer_vocab=<vocab.txt> '
                                      '(or set LDETR_BERT_VOCAB), or pass TextTokens / TextFeatures')
        if module.text_mode == 'features':
            raise NotImplementedError("bbox_text strings need the text encoder: construct the module with text_mode='encoder' (or 'encoder+lm')")
        from .tokenizer import texts_to_tokens
        return texts_to_tokens(module.tokenizer, bbox_text, module.max_text_length, device)
    return bbox_text


def _text_inputs(module, bbox_text, B, N, device):
    if isinstance(bbox_text, TextFeatures):
        return bbox_text.text_feat.to(device=device, dtype=torch.float32), bbox_text.text_len.to(device=device, dtype=torch.int64)
    if isinstance(bbox_text, TextTokens):
        T = bbox_text.input_ids.shape[-1]
        out = module.text_encoder(bbox_text.input_ids.reshape(B * N, T).to(device), attention_mask=bbox_text.attention_mask.reshape(B * N, T).to(device),
                                  return_dict=True, mode='text')
        return out.last_hidden_state[:, 0, :].reshape(B, N, -1), bbox_text.text_len.to(device=device, dtype=torch.int64)
    raise NotImplementedError('pass TextFeatures (precomputed CLS features) or TextTokens (tokenizer output); strings need the host tokenizer')


def _check_background_size(background_size):
    """Up to 256 memory tokens (a 512 x 512 background, stride 32) the fused attention kernels keep a sample's whole score row in
    registers; above that (the reference's signature default 1024 -> 1024 tokens) they walk the keys in chunks of 256 with an online
    softmax (csrc/attention.hip attn_fwd_long_kernel).  The C ABI bounds Lk at 16384 tokens = 4096 x 4096 pixels."""
    if ((background_size + 31) // 32) ** 2 > 16384:
        raise NotImplementedError(f'background_size={background_size}: at most 4096 (16384 image tokens)')


def _masked_ce_static(logits, target, valid):
    """mean cross entropy over the valid slots of full-slot logits [B, N, L] (static-shape heads): padded slots are ignored targets of the
    fused softmax-cross-entropy kernel."""
    from .med import softmax_cross_entropy
    return softmax_cross_entropy(logits.flatten(0, 1), target.flatten().masked_fill(~valid.flatten(), -100))


def _zero_like_loss(ref):
    return ref.new_full((), 0.0)   # a fill kernel: new_zeros(()) becomes a 4-byte memset node under capture (see DESIGN §6 on memset nodes)


class Generator(nn.Module):
    def __init__(self, z_dim, num_bbox_labels, img_channels, img_height, img_width, c_dim,
                 f_dim=256, num_heads=4, num_layers=8, hidden_dim=256,
                 med_config='configs/med_config.json', bert_f_dim=768, bert_num_encoder_layers=12, bert_num_decoder_layers=12,
                 bert_num_heads=12, background_size=1024, im_f_dim=512, max_text_length=256, text_mode=None, tokenizer_vocab=None):
        super().__init__()
        self.z_dim = z_dim
        self.tokenizer = _build_tokenizer(tokenizer_vocab)
        self.num_bbox_labels = num_bbox_labels
        self.c_dim = c_dim
        self.max_text_length = max_text_length
        self.text_mode = text_mode = _resolve_text_mode(text_mode, self.tokenizer)
        self.static_shapes = False   # see module docstring: True = sync-free full-slot outputs (hipGraph-capturable)
        _check_background_size(background_size)

        self.backbone = build_backbone()
        self.input_proj = Conv1x1(self.backbone.num_channels, hidden_dim, kernel_size=1)
        self.fc_z = Linear(z_dim * 9, bert_f_dim)
        self.emb_label = Embedding(num_bbox_labels, bert_f_dim)
        self.text_encoder = _build_text_encoder(text_mode, med_config, bert_num_encoder_layers, bert_num_heads)
        self.text_decoder = _build_text_decoder(text_mode, med_config, bert_num_decoder_layers, bert_num_heads, im_f_dim)
        self.enc_text_len = Embedding(max_text_length, bert_f_dim)
        self.fc_in = MLP(input_dim=4 * bert_f_dim, hidden_dim=bert_f_dim, output_dim=hidden_dim, num_layers=3)
        self.transformer = Transformer(d_model=hidden_dim, dropout=0.1, nhead=8, dim_feedforward=2048, num_encoder_layers=6,
                                       num_decoder_layers=6, normalize_before=False, return_intermediate_dec=False)
        self.bbox_embed = MLP(input_dim=hidden_dim, hidden_dim=hidden_dim, output_dim=4, num_layers=3)
        # reconstructor heads
        self.fc_z_rec = Linear(hidden_dim, z_dim * 9)
        self.fc_out_cls = Linear(hidden_dim, num_bbox_labels)
        self.fc_text_len_rec = Linear(hidden_dim, max_text_length)

    def forward(self, z, bbox_class, bbox_real, bbox_text, bbox_patch, padding_mask, background, c, reconst=False):
        if isinstance(background, (list, torch.Tensor)):
            background = nested_tensor_from_tensor_list(background)
        bg_feat, pos = self.backbone(background)
        bg_feat, mask = bg_feat[-1].decompose()
        assert mask is not None

        B, N = bbox_patch.shape[0], bbox_patch.shape[1]
        bbox_text = _coerce_text(self, bbox_text, bbox_class.device)
        z0 = normalize_2nd_moment(z.view(B, -1))
        zf = self.fc_z(z0).unsqueeze(1).expand(-1, N, -1)
        l = self.emb_label(bbox_class)
        text_feat, text_len = _text_inputs(self, bbox_text, B, N, bbox_class.device)
        text_len_feat = self.enc_text_len(text_len)
        x = torch.cat([zf, l, text_feat, text_len_feat], dim=-1)
        x = self.fc_in(x, final_relu=True).permute(1, 0, 2)

        x = self.transformer(src=self.input_proj(bg_feat), mask=mask, pos_embed=pos[-1], tgt=x, tgt_key_padding_mask=padding_mask)[0]
        bbox_fake = self.bbox_embed(x).sigmoid()
        if not reconst:
            return bbox_fake

        valid = ~padding_mask
        if self.static_shapes:
            z_rec = self.fc_z_rec(x)
            if z_rec.is_cuda and not z0.requires_grad:
                from ..hip.losses import masked_mse
                loss_z = masked_mse(z_rec, z0, valid.contiguous().view(torch.uint8), bdiv=N)      # F.mse_loss(z_rec[valid], z0 per sample): one launch per direction
            else:
                vf = valid.to(torch.float32)
                cnt = vf.sum().clamp_min(1.0)
                loss_z = ((z_rec - z0.unsqueeze(1)).square().sum(-1) * vf).sum() / (cnt * z_rec.shape[-1])
            logit_cls = self.fc_out_cls(x)                                   # [B, N, L]: every slot, masked by the caller
            loss_text_len = _masked_ce_static(self.fc_text_len_rec(x), text_len, valid)
            loss_lm = _lm_loss(self, bbox_text, padding_mask, B, N, True)
            return bbox_fake, loss_z, logit_cls, (loss_lm if loss_lm is not None else _zero_like_loss(loss_z)), loss_text_len
        xv = x[valid]
        z_rec = self.fc_z_rec(xv)
        loss_z = F.mse_loss(z_rec, z0.unsqueeze(1).expand(-1, N, -1)[valid])
        logit_cls = self.fc_out_cls(xv)
        loss_lm = _lm_loss(self, bbox_text, padding_mask, B, N, False)
        loss_lm = loss_lm if loss_lm is not None else _zero_like_loss(loss_z)
        text_len_rec = self.fc_text_len_rec(xv)
        loss_text_len = F.cross_entropy(text_len_rec, text_len[valid])
        return bbox_fake, loss_z, logit_cls, loss_lm, loss_text_len


class Discriminator(nn.Module):
    def __init__(self, num_bbox_labels, img_channels, img_height, img_width, c_dim,
                 f_dim=256, num_heads=4, num_layers=8, max_bbox=50, hidden_dim=256,
                 med_config='configs/med_config.json', bert_f_dim=768, bert_num_encoder_layers=12, bert_num_decoder_layers=12,
                 bert_num_heads=12, background_size=1024, im_f_dim=512, max_text_length=256, text_mode=None, tokenizer_vocab=None):
        super().__init__()
        self.tokenizer = _build_tokenizer(tokenizer_vocab)
        self.num_bbox_labels = num_bbox_labels
        self.c_dim = c_dim
        self.max_text_length = max_text_length
        self.text_mode = text_mode = _resolve_text_mode(text_mode, self.tokenizer)
        self.static_shapes = False
        _check_background_size(background_size)

        # encoder
        self.backbone = build_backbone()
        self.input_proj = Conv1x1(self.backbone.num_channels, hidden_dim, kernel_size=1)
        self.fc_bbox = Linear(4, bert_f_dim)
        self.emb_label = Embedding(num_bbox_labels, bert_f_dim)
        self.text_encoder = _build_text_encoder(text_mode, med_config, bert_num_encoder_layers, bert_num_heads)
        self.text_decoder = _build_text_decoder(text_mode, med_config, bert_num_decoder_layers, bert_num_heads, im_f_dim)
        self.enc_text_len = Embedding(max_text_length, bert_f_dim)
        self.enc_fc_in = MLP(input_dim=4 * bert_f_dim, hidden_dim=bert_f_dim, output_dim=hidden_dim, num_layers=3)
        self.enc_transformer = TransformerWithToken(d_model=hidden_dim, dropout=0.1, nhead=8, dim_feedforward=2048,
                                                    num_encoder_layers=6, num_decoder_layers=6, normalize_before=False,
                                                    return_intermediate_dec=False)
        self.fc_out_disc = Linear(hidden_dim, 1)

        # decoder
        self.pos_token = nn.Parameter(torch.rand(max_bbox, 1, hidden_dim))
        self.dec_fc_in = Linear(hidden_dim + hidden_dim, hidden_dim)
        self.dec_transformer = TransformerEncoder(TransformerEncoderLayer(d_model=hidden_dim, nhead=8, dim_feedforward=2048), num_layers=6)
        self.bbox_embed = Linear(hidden_dim, 4)
        self.fc_out_cls = Linear(hidden_dim, num_bbox_labels)
        self.fc_text_len_rec = Linear(hidden_dim, max_text_length)
        self.bg_decoder = Decoder(z_dim=hidden_dim, w_dim=im_f_dim, channel_max=im_f_dim, channel_base=8192, img_channels=img_channels,
                                  img_resolution=background_size, use_noise=False, num_fp16_res=0, conv_clamp=None,
                                  fused_modconv_default=False)

        # unconditional discriminator
        self.fc_bbox_uncond = Linear(4, bert_f_dim)
        self.emb_label_uncond = Embedding(num_bbox_labels, bert_f_dim)
        self.enc_fc_in_uncond = MLP(input_dim=2 * bert_f_dim, hidden_dim=bert_f_dim, output_dim=hidden_dim, num_layers=3)
        self.enc_transformer_uncond = TransformerWithToken_layoutganpp(d_model=hidden_dim, dim_feedforward=2048, nhead=8, num_layers=6)
        self.fc_out_disc_uncond = Linear(hidden_dim, 1)
        self.pos_token_uncond = nn.Parameter(torch.rand(max_bbox, 1, hidden_dim))
        self.dec_fc_in_uncond = Linear(hidden_dim + hidden_dim, hidden_dim)
        self.dec_transformer_uncond = TransformerEncoder(TransformerEncoderLayer(d_model=hidden_dim, nhead=8, dim_feedforward=2048), num_layers=6)
        self.bbox_embed_uncond = Linear(hidden_dim, 4)
        self.fc_out_cls_uncond = Linear(hidden_dim, num_bbox_labels)

    def trunk(self, background):
        """ResNet trunk + position encoding of the backgrounds (reference: the first statement of D.forward,
        training/networks_detr.py:382-385).  Deterministic in train mode (FrozenBatchNorm, no dropout), so one
        evaluation can serve every D pass of a phase that sees the same backgrounds (`trunk_out=`)."""
        if isinstance(background, (list, torch.Tensor)):
            background = nested_tensor_from_tensor_list(background)
        return self.backbone(background)

    def _logits(self, bbox, l, text_feat, text_len_feat, l_uncond, padding_mask, src, mask, pos):
        """The two discriminator scores of a batch of layouts: -> (x0, logit, x0_uncond, logit_uncond)."""
        b = self.fc_bbox(bbox)
        x = torch.cat([b, l, text_feat, text_len_feat], dim=-1)
        x = self.enc_fc_in(x, final_relu=True).permute(1, 0, 2)
        x_uncond = torch.cat([self.fc_bbox_uncond(bbox), l_uncond], dim=-1)
        x_uncond = self.enc_fc_in_uncond(x_uncond, final_relu=True).permute(1, 0, 2)
        # The unconditional encoder (networks_detr.py:243) does not depend on the conditional path: it advances in lock-step with the layout decoder
        # of enc_transformer, one launch per sub-block step for both stacks (hip.stacks); the reference runs them one after the other.
        partner = None if composite.active() else self.enc_transformer_uncond.as_prog(x_uncond, padding_mask)
        if partner is not None:
            x, _, y_uncond = self.enc_transformer(src=src, mask=mask, pos_embed=pos, tgt=x, tgt_key_padding_mask=padding_mask, partner=partner[0])
            x_uncond = partner[1](y_uncond)
        else:
            x = self.enc_transformer(src=src, mask=mask, pos_embed=pos, tgt=x, tgt_key_padding_mask=padding_mask)[0]
            x_uncond = self.enc_transformer_uncond(x_uncond, src_key_padding_mask=padding_mask)
        x0 = x.transpose(0, 1)[0]
        logit_disc = self.fc_out_disc(x0).squeeze(-1)
        x0_uncond = x_uncond[0]
        return x0, logit_disc, x0_uncond, self.fc_out_disc_uncond(x0_uncond).squeeze(-1)

    def _reconstruct(self, x0, x0_uncond, bbox_text, text_len, padding_mask, B, N):
        """The reconstruction heads on the layout token(s) (reference networks_detr.py:312-359)."""
        valid = ~padding_mask
        x = x0.unsqueeze(0).expand(N, -1, -1)
        t = self.pos_token[:N].expand(-1, B, -1)
        x = self.dec_fc_in(torch.cat([x, t], dim=-1), relu=True)
        x_uncond = x0_uncond.unsqueeze(0).expand(N, -1, -1)
        t_uncond = self.pos_token_uncond[:N].expand(-1, B, -1)
        x_uncond = self.dec_fc_in_uncond(torch.cat([x_uncond, t_uncond], dim=-1), relu=True)
        # the conditional and the unconditional reconstruction decoder (networks_detr.py:269, 275-276) are independent stacks of the same geometry:
        # one launch per sub-block step for both
        x, x_uncond = encode_seq_first_pair(self.dec_transformer, x, self.dec_transformer_uncond, x_uncond, padding_mask)
        static = self.static_shapes
        x = x.permute(1, 0, 2) if static else x.permute(1, 0, 2)[valid]
        bbox_pred = self.bbox_embed(x).sigmoid()
        logit_cls = self.fc_out_cls(x)
        text_len_rec = self.fc_text_len_rec(x)
        if static:
            loss_text_len = _masked_ce_static(text_len_rec, text_len, valid)
        else:
            loss_text_len = F.cross_entropy(text_len_rec, text_len[valid])
        loss_lm = _lm_loss(self, bbox_text, padding_mask, B, N, static)
        loss_lm = loss_lm if loss_lm is not None else _zero_like_loss(loss_text_len)
        bg_rec = self.bg_decoder(x0)

        x_uncond = x_uncond.permute(1, 0, 2) if static else x_uncond.permute(1, 0, 2)[valid]
        bbox_pred_uncond = self.bbox_embed_uncond(x_uncond).sigmoid()
        logit_cls_uncond = self.fc_out_cls_uncond(x_uncond)
        return bbox_pred, logit_cls, loss_lm, loss_text_len, bg_rec, bbox_pred_uncond, logit_cls_uncond

    def forward(self, bbox, bbox_class, bbox_text, bbox_patch, padding_mask, background, c, reconst=False, trunk_out=None):
        bg_feat, pos = self.trunk(background) if trunk_out is None else trunk_out
        bg_feat, mask = bg_feat[-1].decompose()
        assert mask is not None

        B, N = bbox_patch.shape[0], bbox_patch.shape[1]
        bbox_text = _coerce_text(self, bbox_text, bbox_class.device)
        l = self.emb_label(bbox_class)
        text_feat, text_len = _text_inputs(self, bbox_text, B, N, bbox_class.device)
        text_len_feat = self.enc_text_len(text_len)
        x0, logit_disc, x0_uncond, logit_disc_uncond = self._logits(bbox, l, text_feat, text_len_feat, self.emb_label_uncond(bbox_class), padding_mask,
                                                                    self.input_proj(bg_feat), mask, pos[-1])
        if not reconst:
            return logit_disc, logit_disc_uncond
        return (logit_disc, logit_disc_uncond) + self._reconstruct(x0, x0_uncond, bbox_text, text_len, padding_mask, B, N)

    def forward_pair(self, bbox_fake, bbox_real, bbox_class, bbox_text, bbox_patch, padding_mask, background, c, trunk_out=None):
        """D(bbox_fake) and D(bbox_real, reconst=True) of the SAME layouts' conditions in one pass — what phase Dmain evaluates with two
        calls (training/loss.py:149,165).  The samples of a batch are independent (FrozenBatchNorm, no batch statistics), so the two
        score paths run as ONE batch of 2B layouts: trunk, input_proj, label / text embeddings and the text encoder are evaluated
        once, every transformer launch covers both halves, and the reconstruction heads see the real half only.  Values equal the
        two separate calls (dropout draws aside).  -> ((logit, logit_uncond) of the fake half, the 9-tuple of the real half)."""
        bg_feat, pos = self.trunk(background) if trunk_out is None else trunk_out
        bg_feat, mask = bg_feat[-1].decompose()
        B, N = bbox_patch.shape[0], bbox_patch.shape[1]
        bbox_text = _coerce_text(self, bbox_text, bbox_class.device)
        l = self.emb_label(bbox_class)
        text_feat, text_len = _text_inputs(self, bbox_text, B, N, bbox_class.device)
        text_len_feat = self.enc_text_len(text_len)
        l_uncond = self.emb_label_uncond(bbox_class)
        two = lambda t: torch.cat([t, t], dim=0)
        x0, logit, x0_uncond, logit_uncond = self._logits(torch.cat([bbox_fake, bbox_real], dim=0), two(l), two(text_feat), two(text_len_feat), two(l_uncond),
                                                            two(padding_mask), two(self.input_proj(bg_feat)), two(mask), two(pos[-1]))
        rec = self._reconstruct(x0[B:], x0_uncond[B:], bbox_text, text_len, padding_mask, B, N)
        return (logit[:B], logit_uncond[:B]), (logit[B:], logit_uncond[B:]) + rec
