"""ORACLE (test infrastructure only — never imported by the product path).

CPU fp32 restatement of `training/networks_detr.py` Generator.forward (:133-187) and Discriminator.forward
(:279-361) as pure functions over a state dict, in hot-path-only mode (BASELINE.md variant A): the frozen
BERT text encoder output `text_feat` [B,N,768] and the character counts `text_len` [B,N] are inputs, and the
LM-decoder loss is 0 (SURVEY §8a rows a16/a17 are boundary inputs).  Dropout = identity (eval parity).
PINNED: tests/golden/composition.npz holds the outputs of the reference's own Generator.forward / Discriminator.forward
(driven by oracle/gen_golden.py:gen_composition at the real layer sizes, with stand-ins only for the torchvision ResNet-50
body, the BERT tokenizer/encoder and the LM decoder); tests/test_oracle_golden.py checks these functions against it.
Only the torchvision ResNet-50 restatement (detr_ref.resnet50_layer4) stays unpinned (torchvision is absent here).
"""
import torch
import torch.nn.functional as F

from . import detr_ref, stylegan2_ref

NHEAD = 8


def normalize_2nd_moment(x, eps=1e-8):
    return x * (x.square().mean(dim=1, keepdim=True) + eps).rsqrt()


def _lin(sd, pre, x):
    return F.linear(x, sd[pre + 'weight'], sd[pre + 'bias'])


def _backbone(sd, pre, background, feats=None):
    """Joiner(Backbone, PositionEmbeddingSine) -> (input_proj(feat), mask, pos).
    `background`: a [B,3,H,W] tensor (uniform canvas: all-False mask) or a list of [3,Hi,Wi] tensors, zero-padded to the largest
    with mask = True on the padding (detr_util/misc.py:315-337), the mask resized to the feature map by nearest-neighbour
    interpolation (detr_backbone.py:87-92).  `feats` replaces the ResNet-50 body's output (the composition fixtures pin everything
    around the torchvision body, which does not exist in the build container)."""
    if isinstance(background, (list, tuple)):
        C = background[0].shape[0]
        H = max(t.shape[1] for t in background); W = max(t.shape[2] for t in background)
        canvas = torch.zeros(len(background), C, H, W, dtype=background[0].dtype)
        m = torch.ones(len(background), H, W, dtype=torch.bool)
        for i, t in enumerate(background):
            canvas[i, :, :t.shape[1], :t.shape[2]] = t
            m[i, :t.shape[1], :t.shape[2]] = False
    else:
        canvas = background
        m = torch.zeros(background.shape[0], background.shape[2], background.shape[3], dtype=torch.bool)
    feat = detr_ref.resnet50_layer4(sd, pre + 'backbone.0.body.', canvas) if feats is None else feats
    B, _, h, w = feat.shape
    mask = F.interpolate(m[None].float(), size=(h, w)).to(torch.bool)[0]
    pos = detr_ref.position_embedding_sine(mask, 128)
    src = F.conv2d(feat, sd[pre + 'input_proj.weight'], sd[pre + 'input_proj.bias'])
    return src, mask, pos


def generator(sd, z, bbox_class, text_feat, text_len, padding_mask, background, reconst=False, feats=None):
    B, N = bbox_class.shape
    src, mask, pos = _backbone(sd, '', background, feats)
    z0 = normalize_2nd_moment(z.reshape(B, -1))
    zf = _lin(sd, 'fc_z.', z0).unsqueeze(1).expand(-1, N, -1)
    l = sd['emb_label.weight'][bbox_class]
    tl = sd['enc_text_len.weight'][text_len]
    x = torch.cat([zf, l, text_feat, tl], -1)
    x = torch.relu(detr_ref.mlp(sd, 'fc_in.', x, 3)).permute(1, 0, 2)
    x = detr_ref.transformer(sd, src, mask, pos, x, padding_mask, NHEAD, pre='transformer.')[0]
    bbox_fake = detr_ref.mlp(sd, 'bbox_embed.', x, 3).sigmoid()
    if not reconst:
        return bbox_fake
    valid = ~padding_mask
    xv = x[valid]
    loss_z = F.mse_loss(_lin(sd, 'fc_z_rec.', xv), z0.unsqueeze(1).expand(-1, N, -1)[valid])
    logit_cls = _lin(sd, 'fc_out_cls.', xv)
    loss_lm = loss_z.new_zeros(())
    loss_text_len = F.cross_entropy(_lin(sd, 'fc_text_len_rec.', xv), text_len[valid])
    return bbox_fake, loss_z, logit_cls, loss_lm, loss_text_len


def discriminator(sd, bbox, bbox_class, text_feat, text_len, padding_mask, background, reconst=False, bg_size=256, feats=None):
    B, N = bbox_class.shape
    src, mask, pos = _backbone(sd, '', background, feats)
    b = _lin(sd, 'fc_bbox.', bbox)
    l = sd['emb_label.weight'][bbox_class]
    tl = sd['enc_text_len.weight'][text_len]
    x = torch.cat([b, l, text_feat, tl], -1)
    x = torch.relu(detr_ref.mlp(sd, 'enc_fc_in.', x, 3)).permute(1, 0, 2)
    x = detr_ref.transformer(sd, src, mask, pos, x, padding_mask, NHEAD, pre='enc_transformer.', with_token=True)[0].transpose(0, 1)
    x0 = x[0]
    logit = _lin(sd, 'fc_out_disc.', x0).squeeze(-1)

    xu = torch.cat([_lin(sd, 'fc_bbox_uncond.', bbox), sd['emb_label_uncond.weight'][bbox_class]], -1)
    xu = torch.relu(detr_ref.mlp(sd, 'enc_fc_in_uncond.', xu, 3)).permute(1, 0, 2)
    xu = detr_ref.token_encoder_layoutganpp(sd, 'enc_transformer_uncond.', xu, padding_mask, NHEAD)
    x0u = xu[0]
    logit_u = _lin(sd, 'fc_out_disc_uncond.', x0u).squeeze(-1)
    if not reconst:
        return logit, logit_u

    valid = ~padding_mask
    x = torch.cat([x0.unsqueeze(0).expand(N, -1, -1), sd['pos_token'][:N].expand(-1, B, -1)], -1)
    x = torch.relu(_lin(sd, 'dec_fc_in.', x))
    x = detr_ref.torch_encoder(sd, 'dec_transformer.', x, NHEAD, padding_mask).permute(1, 0, 2)[valid]
    bbox_pred = _lin(sd, 'bbox_embed.', x).sigmoid()
    logit_cls = _lin(sd, 'fc_out_cls.', x)
    loss_text_len = F.cross_entropy(_lin(sd, 'fc_text_len_rec.', x), text_len[valid])
    loss_lm = loss_text_len.new_zeros(())
    bg_rec = stylegan2_ref.decoder(sd, 'bg_decoder.', x0, bg_size)

    xu = torch.cat([x0u.unsqueeze(0).expand(N, -1, -1), sd['pos_token_uncond'][:N].expand(-1, B, -1)], -1)
    xu = torch.relu(_lin(sd, 'dec_fc_in_uncond.', xu))
    xu = detr_ref.torch_encoder(sd, 'dec_transformer_uncond.', xu, NHEAD, padding_mask).permute(1, 0, 2)[valid]
    bbox_pred_u = _lin(sd, 'bbox_embed_uncond.', xu).sigmoid()
    logit_cls_u = _lin(sd, 'fc_out_cls_uncond.', xu)
    return logit, logit_u, bbox_pred, logit_cls, loss_lm, loss_text_len, bg_rec, bbox_pred_u, logit_cls_u
