"""ORACLE (test infrastructure only — never imported by the product path).

CPU fp32 restatement of the DETR pieces of the hot path, written as pure functions over a state dict
(keys are the reference's parameter names) so it shares no code with layoutdetr_amd:
  multi-head attention as nn.MultiheadAttention computes it (packed in_proj, q scaled by 1/sqrt(dh),
    boolean key_padding_mask -> -inf, softmax, out_proj)  — called at training/detr_transformer.py:208-209,273-280
  TransformerEncoderLayer.forward_post   training/detr_transformer.py:202-215
  TransformerDecoderLayer.forward_post   training/detr_transformer.py:265-286
  Transformer / TransformerWithToken     training/detr_transformer.py:55-70, 102-112
  nn.TransformerEncoder branch           training/util.py:13-43 (post-norm, relu, eps 1e-5)
  PositionEmbeddingSine                  training/detr_position_encoding.py:38-58
  FrozenBatchNorm2d                      training/detr_backbone.py:55-65
  ResNet-50 v1.5 trunk                   torchvision==0.13.1 resnet50 (third-party; absent here: restated from
                                         its published architecture; PARITY UNPINNED by any reference vector)
Dropout is the identity here (eval-mode parity; see DESIGN.md for the statistical dropout checks).
Pinned by tests/test_oracle_golden.py against tests/golden/{transformer,transformer_token,
transformer_layoutganpp,pos_encoding,frozen_bn}.npz.
"""
import math

import torch
import torch.nn.functional as F


def mha(sd, pre, query, key, value, nhead, key_padding_mask=None):
    """Seq-first [L, B, E] tensors, like the reference."""
    Lq, B, E = query.shape
    Lk = key.shape[0]
    dh = E // nhead
    W, b = sd[pre + 'in_proj_weight'], sd[pre + 'in_proj_bias']
    q = F.linear(query, W[:E], b[:E]) * (1.0 / math.sqrt(dh))
    k = F.linear(key, W[E:2 * E], b[E:2 * E])
    v = F.linear(value, W[2 * E:], b[2 * E:])
    q = q.reshape(Lq, B * nhead, dh).transpose(0, 1)
    k = k.reshape(Lk, B * nhead, dh).transpose(0, 1)
    v = v.reshape(Lk, B * nhead, dh).transpose(0, 1)
    s = torch.bmm(q, k.transpose(1, 2))
    if key_padding_mask is not None:
        s = s.view(B, nhead, Lq, Lk).masked_fill(key_padding_mask[:, None, None, :], float('-inf')).view(B * nhead, Lq, Lk)
    p = s.softmax(-1)
    o = torch.bmm(p, v).transpose(0, 1).reshape(Lq, B, E)
    return F.linear(o, sd[pre + 'out_proj.weight'], sd[pre + 'out_proj.bias'])


def _ln(sd, pre, x):
    return F.layer_norm(x, (x.shape[-1],), sd[pre + 'weight'], sd[pre + 'bias'], 1e-5)


def _ffn(sd, pre, x):
    h = F.relu(F.linear(x, sd[pre + 'linear1.weight'], sd[pre + 'linear1.bias']))
    return F.linear(h, sd[pre + 'linear2.weight'], sd[pre + 'linear2.bias'])


def encoder_layer(sd, pre, src, nhead, key_padding_mask, pos):
    qk = src if pos is None else src + pos
    src = _ln(sd, pre + 'norm1.', src + mha(sd, pre + 'self_attn.', qk, qk, src, nhead, key_padding_mask))
    return _ln(sd, pre + 'norm2.', src + _ffn(sd, pre, src))


def decoder_layer(sd, pre, tgt, memory, nhead, tgt_kpm, mem_kpm, pos):
    tgt = _ln(sd, pre + 'norm1.', tgt + mha(sd, pre + 'self_attn.', tgt, tgt, tgt, nhead, tgt_kpm))
    tgt = _ln(sd, pre + 'norm2.', tgt + mha(sd, pre + 'multihead_attn.', tgt, memory + pos, memory, nhead, mem_kpm))
    return _ln(sd, pre + 'norm3.', tgt + _ffn(sd, pre, tgt))


def _count_layers(sd, pre):
    n = 0
    while (pre + f'{n}.norm1.weight') in sd:
        n += 1
    return n


def transformer(sd, src, mask, pos_embed, tgt, tgt_key_padding_mask, nhead, pre='', with_token=False):
    """Returns (hs [B, Lq(+1), C], memory [B, C, h, w])."""
    bs, c, h, w = src.shape
    src = src.flatten(2).permute(2, 0, 1)
    pos = pos_embed.flatten(2).permute(2, 0, 1)
    m = mask.flatten(1)
    x = src
    for i in range(_count_layers(sd, pre + 'encoder.layers.')):
        x = encoder_layer(sd, pre + f'encoder.layers.{i}.', x, nhead, m, pos)
    memory = x
    if with_token:
        tgt = torch.cat([sd[pre + 'token'].expand(-1, bs, -1), tgt], 0)
        tgt_key_padding_mask = torch.cat([sd[pre + 'token_mask'].expand(bs, -1), tgt_key_padding_mask], 1)
    y = tgt
    for i in range(_count_layers(sd, pre + 'decoder.layers.')):
        y = decoder_layer(sd, pre + f'decoder.layers.{i}.', y, memory, nhead, tgt_key_padding_mask, m, pos)
    y = _ln(sd, pre + 'decoder.norm.', y)
    return y.transpose(0, 1), memory.permute(1, 2, 0).reshape(bs, c, h, w)


def torch_encoder(sd, pre, x, nhead, key_padding_mask):
    """nn.TransformerEncoder of post-norm nn.TransformerEncoderLayer(relu); x seq-first [L, B, E]."""
    for i in range(_count_layers(sd, pre + 'layers.')):
        x = encoder_layer(sd, pre + f'layers.{i}.', x, nhead, key_padding_mask, None)
    return x


def token_encoder_layoutganpp(sd, pre, x, src_key_padding_mask, nhead):
    """training/util.py:28-43."""
    B = x.shape[1]
    x = torch.cat([sd[pre + 'token'].expand(-1, B, -1), x], 0)
    kpm = torch.cat([sd[pre + 'token_mask'].expand(B, -1), src_key_padding_mask], 1)
    return torch_encoder(sd, pre + 'core.', x, nhead, kpm)


def position_embedding_sine(mask, num_pos_feats=128, temperature=10000.0):
    """mask [B, h, w] bool (True = padded) -> [B, 2*num_pos_feats, h, w]; normalize=True, scale 2*pi."""
    not_mask = ~mask
    y_embed = not_mask.cumsum(1, dtype=torch.float32)
    x_embed = not_mask.cumsum(2, dtype=torch.float32)
    eps, scale = 1e-6, 2 * math.pi
    y_embed = y_embed / (y_embed[:, -1:, :] + eps) * scale
    x_embed = x_embed / (x_embed[:, :, -1:] + eps) * scale
    i = torch.arange(num_pos_feats, dtype=torch.float32)
    dim_t = temperature ** (2 * torch.div(i, 2, rounding_mode='floor') / num_pos_feats)
    px = x_embed[:, :, :, None] / dim_t
    py = y_embed[:, :, :, None] / dim_t
    px = torch.stack((px[..., 0::2].sin(), px[..., 1::2].cos()), dim=4).flatten(3)
    py = torch.stack((py[..., 0::2].sin(), py[..., 1::2].cos()), dim=4).flatten(3)
    return torch.cat((py, px), dim=3).permute(0, 3, 1, 2)


def frozen_bn(sd, pre, x):
    scale = sd[pre + 'weight'] * (sd[pre + 'running_var'] + 1e-5).rsqrt()
    bias = sd[pre + 'bias'] - sd[pre + 'running_mean'] * scale
    return x * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1)


RESNET50_BLOCKS = (3, 4, 6, 3)


def resnet50_layer4(sd, pre, x):
    """torchvision resnet50 (v1.5: stride on the 3x3) trunk with FrozenBatchNorm2d; returns layer4 [B,2048,H/32,W/32]."""
    x = F.conv2d(x, sd[pre + 'conv1.weight'], stride=2, padding=3)
    x = F.relu(frozen_bn(sd, pre + 'bn1.', x))
    x = F.max_pool2d(x, 3, 2, 1)
    for li, nblocks in enumerate(RESNET50_BLOCKS, start=1):
        for bi in range(nblocks):
            p = f'{pre}layer{li}.{bi}.'
            stride = 2 if (bi == 0 and li > 1) else 1
            idt = x
            out = F.relu(frozen_bn(sd, p + 'bn1.', F.conv2d(x, sd[p + 'conv1.weight'])))
            out = F.relu(frozen_bn(sd, p + 'bn2.', F.conv2d(out, sd[p + 'conv2.weight'], stride=stride, padding=1)))
            out = frozen_bn(sd, p + 'bn3.', F.conv2d(out, sd[p + 'conv3.weight']))
            if (p + 'downsample.0.weight') in sd:
                idt = frozen_bn(sd, p + 'downsample.1.', F.conv2d(x, sd[p + 'downsample.0.weight'], stride=stride))
            x = F.relu(out + idt)
    return x


def mlp(sd, pre, x, num_layers):
    """training/networks_detr.py:50-62."""
    for i in range(num_layers):
        x = F.linear(x, sd[pre + f'layers.{i}.weight'], sd[pre + f'layers.{i}.bias'])
        if i < num_layers - 1:
            x = F.relu(x)
    return x
