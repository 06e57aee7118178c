"""ORACLE (test infrastructure only — never imported by the product path).

CPU fp32 restatement of one G/D training iteration in hot-path-only mode:
  StyleGAN2Loss.accumulate_gradients, phases Gmain and Dmain with gamma=0, pl_weight=0
      training/loss.py:84-116 (Gmain), :146-157 (Dgen), :161-218 (Dreal); weights from train.py:263-275
  and the two regulariser phases: Greg (path length, :119-142) and Dreg (R1, :162-166 + :207-217) -- pinned by tests/golden/reg.npz, the
  reference's own accumulate_gradients('Greg' / 'Dreg') in fp32 and fp64 (oracle/gen_golden.py:gen_reg)
  gradient post-processing + Adam(betas=(0,0.99), eps=1e-8)      training/training_loop.py:303-313, train.py:204-205
Built on the pinned pieces (networks_ref / detr_ref / stylegan2_ref / losses_ref), and PINNED as a whole: tests/golden/composition.npz
holds every training_stats-reported term and the parameter gradients of the reference's own StyleGAN2Loss.accumulate_gradients
('Gmain', 'Dmain') at the real layer sizes (oracle/gen_golden.py:gen_composition); tests/test_oracle_golden.py compares.  Used by the GPU parity
tests, by __graft_entry__.smoke() and as bench.py's `cpu_baseline` ("port").
"""
import torch
import torch.nn.functional as F

from . import losses_ref, networks_ref

WEIGHTS = dict(Dreal_bbox_cls=50.0, Dreal_bbox_rec=500.0, Dreal_text_rec=0.1, Dreal_text_len_rec=2.0, Dreal_im_rec=0.5,
               Ggen_bbox_rec=100.0, Ggen_bbox_gIoU=4.0, Ggen_overlapping=7.0, Ggen_alignment=17.0, Ggen_z_rec=5.0,
               Ggen_bbox_cls=50.0, Ggen_text_rec=1.0, Ggen_text_len_rec=1.0)


def g_main_loss(G, D, bt, z, w=WEIGHTS, bg_size=256, terms=None):
    """loss.py:84-116.  `terms` (dict, optional) receives every value the reference hands to training_stats.report."""
    pm = bt['padding_mask']; valid = ~pm
    bbox_fake, loss_z, cls_logits, loss_lm, loss_tl = networks_ref.generator(
        G, z, bt['bbox_class'], bt['text_feat'], bt['text_len'], pm, bt['background'], reconst=True, feats=bt.get('feats_G'))
    logits, logits_u = networks_ref.discriminator(D, bbox_fake, bt['bbox_class'], bt['text_feat'], bt['text_len'], pm, bt['background'],
                                                  feats=bt.get('feats_D'))
    real = bt['bbox_real']
    t = {'Loss/scores/fake': logits, 'Loss/signs/fake': logits.sign(),
         'Loss/G/loss_Ggen': F.softplus(-logits), 'Loss/G/loss_Ggen_uncond': F.softplus(-logits_u),
         'Loss/G/loss_Ggen_bbox_rec': F.mse_loss(bbox_fake[valid], real[valid]) * w['Ggen_bbox_rec'],
         'Loss/G/loss_Ggen_bbox_gIoU': losses_ref.generalized_iou_loss(bbox_fake[valid], real[valid]) * w['Ggen_bbox_gIoU'],
         'Loss/G/loss_Ggen_overlapping': losses_ref.compute_overlap(bbox_fake, valid) * w['Ggen_overlapping'],
         'Loss/G/loss_Ggen_alignment': losses_ref.compute_alignment(bbox_fake, valid) * w['Ggen_alignment'],
         'Loss/G/loss_Ggen_z_rec': loss_z * w['Ggen_z_rec'],
         'Loss/G/loss_Ggen_bbox_cls': F.cross_entropy(cls_logits, bt['bbox_class'][valid]) * w['Ggen_bbox_cls'],
         'Loss/G/loss_Ggen_text_rec': loss_lm * w['Ggen_text_rec'], 'Loss/G/loss_Ggen_text_len_rec': loss_tl * w['Ggen_text_len_rec']}
    if terms is not None:
        terms.update({k: v.detach() for k, v in t.items()})
    total = sum(v for k, v in t.items() if k.startswith('Loss/G/'))
    return total.mean(), bbox_fake


def d_gen_loss(G, D, bt, z, terms=None):
    """loss.py:146-157."""
    pm = bt['padding_mask']
    with torch.no_grad():
        bbox_fake = networks_ref.generator(G, z, bt['bbox_class'], bt['text_feat'], bt['text_len'], pm, bt['background'], feats=bt.get('feats_G'))
    logits, logits_u = networks_ref.discriminator(D, bbox_fake, bt['bbox_class'], bt['text_feat'], bt['text_len'], pm, bt['background'],
                                                  feats=bt.get('feats_D'))
    t = {'Loss/scores/fake': logits, 'Loss/signs/fake': logits.sign(), 'Loss/D/loss_Dgen': F.softplus(logits), 'Loss/D/loss_Dgen_uncond': F.softplus(logits_u)}
    if terms is not None:
        terms.update({k: v.detach() for k, v in t.items()})
    return (t['Loss/D/loss_Dgen'] + t['Loss/D/loss_Dgen_uncond']).mean()


def d_real_loss(D, bt, w=WEIGHTS, bg_size=256, terms=None):
    """loss.py:161-218 (phase Dmain: no R1 term)."""
    pm = bt['padding_mask']; valid = ~pm
    real = bt['bbox_real']
    (logits, logits_u, bbox_rec, cls_logits, loss_lm, loss_tl, bg_rec, bbox_rec_u, cls_logits_u) = networks_ref.discriminator(
        D, real, bt['bbox_class'], bt['text_feat'], bt['text_len'], pm, bt['background'], reconst=True, bg_size=bg_size, feats=bt.get('feats_D'))
    t = {'Loss/scores/real': logits, 'Loss/signs/real': logits.sign(),
         'Loss/D/loss_Dreal': F.softplus(-logits), 'Loss/D/loss_Dreal_uncond': F.softplus(-logits_u),
         'Loss/D/loss_Dreal_bbox_rec': F.mse_loss(bbox_rec, real[valid]) * w['Dreal_bbox_rec'],
         'Loss/D/loss_Dreal_bbox_cls': F.cross_entropy(cls_logits, bt['bbox_class'][valid]) * w['Dreal_bbox_cls'],
         'Loss/D/loss_Dreal_text_rec': loss_lm * w['Dreal_text_rec'], 'Loss/D/loss_Dreal_text_len_rec': loss_tl * w['Dreal_text_len_rec'],
         'Loss/D/loss_Dreal_bg_rec': F.mse_loss(bg_rec, bt['background']) * w['Dreal_im_rec'],
         'Loss/D/loss_Dreal_bbox_rec_uncond': F.mse_loss(bbox_rec_u, real[valid]) * w['Dreal_bbox_rec'],
         'Loss/D/loss_Dreal_bbox_cls_uncond': F.cross_entropy(cls_logits_u, bt['bbox_class'][valid]) * w['Dreal_bbox_cls']}
    if terms is not None:
        terms.update({k: v.detach() for k, v in t.items()})
    return sum(v for k, v in t.items() if k.startswith('Loss/D/')).mean()


def d_r1_loss(D, bt, r1_gamma, terms=None):
    """loss.py:162-166, 207-217 in phase 'Dreg': gamma / 2 * || d D(real).sum() / d bbox_real ||^2 per sample (create_graph: the penalty is
    differentiated again by the caller's backward)."""
    real = bt['bbox_real'].detach().requires_grad_(True)
    logits = networks_ref.discriminator(D, real, bt['bbox_class'], bt['text_feat'], bt['text_len'], bt['padding_mask'], bt['background'],
                                        feats=bt.get('feats_D'))[0]
    r1_grads = torch.autograd.grad([logits.sum()], [real], create_graph=True, only_inputs=True)[0]
    r1_penalty = r1_grads.square().sum([1, 2])
    loss = r1_penalty * (r1_gamma / 2)
    if terms is not None:
        terms.update({'Loss/scores/real': logits.detach(), 'Loss/signs/real': logits.detach().sign(), 'Loss/r1_penalty': r1_penalty.detach(),
                      'Loss/D/reg': loss.detach(), 'r1_grads': r1_grads.detach()})
    return loss.mean()


def g_pl_loss(G, bt, z, noise, pl_mean, pl_weight, pl_batch_shrink=2, pl_decay=0.01, terms=None):
    """loss.py:119-142 (phase 'Greg').  `noise` = the torch.randn_like(bbox_fake) draw of :131; `pl_mean` = the running mean before this call.
    -> (loss, new running mean)."""
    bs = z.shape[0] // pl_batch_shrink
    zt = z[:bs].detach().requires_grad_(True)
    feats = bt.get('feats_G')
    bbox_fake = networks_ref.generator(G, zt, bt['bbox_class'][:bs], bt['text_feat'][:bs], bt['text_len'][:bs], bt['padding_mask'][:bs], bt['background'][:bs],
                                       feats=None if feats is None else feats[:bs])
    pl_noise = noise / float(bbox_fake.shape[2])
    pl_grads = torch.autograd.grad([(bbox_fake * pl_noise).sum()], [zt], create_graph=True, only_inputs=True)[0]
    pl_lengths = pl_grads.square().sum([1, 2]).sqrt()
    new_mean = pl_mean.lerp(pl_lengths.mean(), pl_decay)
    pl_penalty = (pl_lengths - new_mean).square()
    loss = pl_penalty * pl_weight
    if terms is not None:
        terms.update({'Loss/pl_penalty': pl_penalty.detach(), 'Loss/G/reg': loss.detach(), 'pl_grads': pl_grads.detach(), 'pl_mean': new_mean.detach()})
    return loss.mean(), new_mean.detach()


def _params(sd, param_names=None):
    """state dict -> leaf tensors; `param_names` (names of nn.Parameters) decides what is trainable, buffers
    (FrozenBatchNorm statistics, FIR taps, token masks, w_avg) never are."""
    out = {}
    for k, v in sd.items():
        if param_names is not None:
            trainable = k in param_names
        else:
            trainable = v.dtype.is_floating_point and not any(s in k for s in (
                'running_mean', 'running_var', '.bn', 'token_mask', 'downsample.1.', 'w_avg', 'resample_filter'))
        out[k] = v.detach().clone().requires_grad_(bool(trainable))
    return out


def training_iteration(G_sd, D_sd, bt, z_g, z_d, lr=1e-5, world=1, bg_size=256, apply_adam=True, G_param_names=None, D_param_names=None):
    """One Gmain + Dmain iteration on CPU.  Returns (losses dict, grads dicts, updated state dicts)."""
    G = _params(G_sd, G_param_names); D = _params(D_sd, D_param_names)
    out = {}
    # Gmain: D is frozen (its parameters get no gradient), gradient flows through D to bbox_fake.
    Dfrozen = {k: v.detach() for k, v in D.items()}
    tG, tD = {}, {}
    btG = dict(bt, **({'feats_D': bt['feats_D'].detach()} if 'feats_D' in bt else {}))     # D (and its trunk output) is frozen in Gmain
    btD = dict(bt, **({'feats_G': bt['feats_G'].detach()} if 'feats_G' in bt else {}))
    lg, bbox_fake = g_main_loss(G, Dfrozen, btG, z_g, bg_size=bg_size, terms=tG)
    lg.backward()
    out['loss_G'] = lg.detach(); out['bbox_fake'] = bbox_fake.detach(); out['terms_G'] = tG; out['terms_D'] = tD
    gG = {k: v.grad for k, v in G.items() if v.requires_grad and v.grad is not None}
    G_new = dict(G_sd)
    if apply_adam:
        G_new = adam_step(G, gG, lr, world)
    # Dmain uses the *updated* generator (phases run sequentially within an iteration).
    Gd = {k: v.detach() for k, v in (G_new if apply_adam else G).items()}
    ld1 = d_gen_loss(Gd, D, btD, z_d, terms=tD); ld1.backward()
    ld2 = d_real_loss(D, btD, bg_size=bg_size, terms=tD); ld2.backward()
    out['loss_Dgen'] = ld1.detach(); out['loss_Dreal'] = ld2.detach()
    gD = {k: v.grad for k, v in D.items() if v.requires_grad and v.grad is not None}
    D_new = adam_step(D, gD, lr, world) if apply_adam else dict(D_sd)
    return out, gG, gD, G_new, D_new


def adam_step(params, grads, lr, world=1, step=1, betas=(0.0, 0.99), eps=1e-8):
    """First Adam step from zero state after the DP post-processing; returns a new state dict."""
    new = {}
    for k, p in params.items():
        if k not in grads:
            new[k] = p.detach()
            continue
        g = losses_ref.dp_postprocess(grads[k] * world, world)
        m = (1 - betas[0]) * g
        v = (1 - betas[1]) * g * g
        bc1 = 1 - betas[0] ** step; bc2 = 1 - betas[1] ** step
        new[k] = p.detach() - (lr / bc1) * m / (v.sqrt() / (bc2 ** 0.5) + eps)
    return new
