"""StyleGAN2Loss for the LayoutDETR G/D step (reference: training/loss.py:28-218) over the gfx950 modules.
Same class name, constructor arguments, phase names and loss composition; the default `gamma=0,
pl_weight=0` configuration (train.py:135-136) makes Greg/Dreg no-ops exactly as loss.py:77-80 does.
R1 (`--gamma`, loss.py:207-215) and path-length regularisation (loss.py:119-142) differentiate D's score / G's boxes with
`create_graph=True`: those phases run the heads and the decoder stack between the differentiated input and output on the
twice-differentiable nodes of hip/composite.py (SURVEY §7 'second-order autograd'); everything else keeps the fused kernels."""

import torch
import torch.nn.functional as F

from ..hip import composite, core

from ..hip import losses as hl
from ..metrics.metric_layoutnet import compute_alignment, compute_overlap, generalized_iou_loss, layout_losses_fused, layout_losses_per_sample


def _masked_mse(a, b, valid):
    """F.mse_loss(a[valid], b[valid]) without the gather: a, b [B,N,D], valid [B,N] bool."""
    if a.is_cuda and a.dtype == torch.float32 and not b.requires_grad:
        return hl.masked_mse(a, b, valid.contiguous().view(torch.uint8))      # one launch per direction (csrc/layout_loss.hip) instead of ~9 + ~12
    vf = valid.to(a.dtype)
    return ((a - b).square().sum(-1) * vf).sum() / (vf.sum().clamp_min(1.0) * a.shape[-1])


def _masked_ce(logits, target, valid, raw=False):
    """F.cross_entropy(logits[valid], target[valid]) without the gather: logits [B,N,L].  On the GPU the padded slots become ignored
    targets of the fused softmax-cross-entropy kernel (csrc/xent.hip: loss and row log-sum-exp in one pass, gradient in one pass):
    the same mean over the same slots in 4 launches instead of ~13.  raw: the (loss sum, count) pair for hip.losses.combine's RATIO term."""
    if logits.is_cuda and logits.dtype == torch.float32:
        from .med import softmax_cross_entropy
        return softmax_cross_entropy(logits.flatten(0, 1), target.flatten().masked_fill(~valid.flatten(), -100), raw=raw)
    assert not raw
    vf = valid.to(logits.dtype).flatten()
    ce = F.cross_entropy(logits.flatten(0, 1), target.flatten(), reduction='none')
    return (ce * vf).sum() / vf.sum().clamp_min(1.0)


def _masked_giou(a, b, valid):
    """generalized_iou_loss(a[valid], b[valid]) without the gather (padded slots are replaced by a unit box first)."""
    unit = a.new_full((4,), 0.5)
    unit[2:] = 1.0                      # [0.5, 0.5, 1, 1] built on device (no host copy: stays hipGraph-capturable)
    v3 = valid.unsqueeze(-1)
    a2 = torch.where(v3, a, unit).flatten(0, 1)
    b2 = torch.where(v3, b, unit).flatten(0, 1)
    l1, t1, r1, b1 = a2[:, 0] - a2[:, 2] / 2, a2[:, 1] - a2[:, 3] / 2, a2[:, 0] + a2[:, 2] / 2, a2[:, 1] + a2[:, 3] / 2
    l2, t2, r2, bb2 = b2[:, 0] - b2[:, 2] / 2, b2[:, 1] - b2[:, 3] / 2, b2[:, 0] + b2[:, 2] / 2, b2[:, 1] + b2[:, 3] / 2
    a_1, a_2 = (r1 - l1) * (b1 - t1), (r2 - l2) * (bb2 - t2)
    lm, rm, tm, bm = torch.maximum(l1, l2), torch.minimum(r1, r2), torch.maximum(t1, t2), torch.minimum(b1, bb2)
    ai = torch.where((lm < rm) & (tm < bm), (rm - lm) * (bm - tm), torch.zeros_like(a_1))
    au = a_1 + a_2 - ai
    ah = (torch.maximum(r1, r2) - torch.minimum(l1, l2)) * (torch.maximum(b1, bb2) - torch.minimum(t1, t2))
    per = 1.0 - (ai / au - (ah - au) / ah)
    vf = valid.to(a.dtype).flatten()
    return (per * vf).sum() / vf.sum().clamp_min(1.0)


class Loss:
    def accumulate_gradients(self, phase, bbox_real, bbox_class, bbox_text, bbox_patch, padding_mask, background, real_c, gen_z, gen_c, gain, cur_nimg):
        raise NotImplementedError()


class StyleGAN2Loss(Loss):
    def __init__(self, device, G, D, augment_pipe=None, r1_gamma=0.0, style_mixing_prob=0, pl_weight=0.0, pl_batch_shrink=2,
                 pl_decay=0.01, pl_no_weight_grad=False, blur_init_sigma=0, blur_fade_kimg=0,
                 Dreal_bbox_cls_weight=50.0, Dreal_bbox_rec_weight=500.0, Dreal_text_rec_weight=0.1, Dreal_text_len_rec_weight=2.0,
                 Dreal_im_rec_weight=0.5, Ggen_bbox_rec_weight=100.0, Ggen_bbox_gIoU_weight=4.0, Ggen_overlapping_weight=7.0,
                 Ggen_alignment_weight=17.0, Ggen_z_rec_weight=5.0, Ggen_bbox_cls_weight=50.0, Ggen_text_rec_weight=1.0,
                 Ggen_text_len_rec_weight=1.0, report_fn=None, share_D_trunk=True):
        super().__init__()
        self.device = device
        self.G = G
        self.D = D
        self.augment_pipe = augment_pipe
        self.r1_gamma = r1_gamma
        self.style_mixing_prob = style_mixing_prob
        self.pl_weight = pl_weight
        self.pl_batch_shrink = pl_batch_shrink
        self.pl_decay = pl_decay
        self.pl_no_weight_grad = pl_no_weight_grad
        self.pl_mean = torch.zeros([], device=device)
        self.pl_noise_fn = None       # tests: callable(bbox_fake) -> the noise of loss.py:131 instead of torch.randn_like
        self.w = dict(Dreal_bbox_cls=Dreal_bbox_cls_weight, Dreal_bbox_rec=Dreal_bbox_rec_weight, Dreal_text_rec=Dreal_text_rec_weight,
                      Dreal_text_len_rec=Dreal_text_len_rec_weight, Dreal_im_rec=Dreal_im_rec_weight, Ggen_bbox_rec=Ggen_bbox_rec_weight,
                      Ggen_bbox_gIoU=Ggen_bbox_gIoU_weight, Ggen_overlapping=Ggen_overlapping_weight, Ggen_alignment=Ggen_alignment_weight,
                      Ggen_z_rec=Ggen_z_rec_weight, Ggen_bbox_cls=Ggen_bbox_cls_weight, Ggen_text_rec=Ggen_text_rec_weight,
                      Ggen_text_len_rec=Ggen_text_len_rec_weight)
        # Dmain evaluates D on the generated and on the real layout of the SAME backgrounds with the SAME weights; D's ResNet
        # trunk is deterministic, so both passes can read one trunk evaluation and its backward runs once on the summed
        # gradient (the reference recomputes it, training/loss.py:176-210: two run_D calls, two backward calls).  Same losses and
        # gradients up to fp32 summation order; share_D_trunk=False restores the reference's call pattern.
        self.share_D_trunk = share_D_trunk
        # with a shared trunk, D(fake) and D(real) of Dmain also run as one batch of 2B (values identical; LDETR_DEBUG="PAIR_D=0" = two calls)
        self.pair_D_passes = bool(share_D_trunk) and core.knob('PAIR_D', 1) != 0
        self.fused_layout_losses = True   # csrc/layout_loss.hip (static-shape path)
        self.fused_loss_tail = core.knob('FUSED_LOSS_TAIL', 1) != 0           # hip.losses.combine: a phase's tail as one launch per direction
        # share_D_trunk='iteration' goes one step further: D's weights do not change between the Gmain and the Dmain phase of one
        # iteration (Gmain updates G only, training_loop.py:281-313), so ONE trunk evaluation per iteration serves D(fake) in Gmain
        # (values only: D is frozen there) and both D passes of Dmain (with its autograd graph).  The iteration driver calls
        # precompute_D_trunk() before the phases; without that call the per-phase behaviour above applies.
        self._trunk_cache = {}
        self._reporting = report_fn is not None   # the sign() statistics cost a launch each: only formed when someone listens
        self.report = report_fn if report_fn is not None else (lambda name, value: None)
        self.last = {}

    @staticmethod
    def _bg_key(background):
        return (background.data_ptr(), tuple(background.shape)) if isinstance(background, torch.Tensor) else id(background)

    def precompute_D_trunk(self, background, stages=None):
        """Evaluate D's trunk on `background` with gradient tracking and park it for this iteration's phases.
        stages: a detr_backbone.BackwardStages that records the trunk's backward cuts (Dmain's staged backward continues from them)."""
        if self.share_D_trunk != 'iteration' or not hasattr(self.D, 'trunk'):
            return
        body = self.D.backbone[0].body
        # G's trunk for the Gmain phase rides along: same backgrounds, same architecture, different weights -> every convolution of the two trunks
        # as one grouped launch (detr_backbone.dual_trunk_forward); G's output is parked on its body and consumed by Gmain's generator forward.
        # Not with a staged backward (its cuts are recorded inside the ordinary forward) and not for ragged backgrounds.
        g_body = self.G.backbone[0].body if hasattr(self.G, 'backbone') else None
        dual = (stages is None and g_body is not None and isinstance(background, torch.Tensor) and core.knob('DUAL_TRUNK', 1) != 0
                and type(g_body) is type(body) and hasattr(body, '_entrance'))
        trunk_params = list(self.D.backbone.parameters()) + (list(self.G.backbone.parameters()) if dual else [])
        was = [p.requires_grad for p in trunk_params]
        for p in trunk_params:
            p.requires_grad_(True)     # the autograd graphs are built now, used by the phases' backward passes (training_loop.py:282 sets it there)
        try:
            if stages is not None:
                body.stages = stages
            with torch.enable_grad():
                if dual:
                    from .detr_backbone import dual_trunk_forward
                    key = (background.data_ptr(), tuple(background.shape))
                    out_g, out_d = dual_trunk_forward(g_body, body, background, background)
                    g_body.injected = dict(g_body.injected or {}); g_body.injected[key] = out_g
                    body.injected = {key: out_d}
                self._trunk_cache[self._bg_key(background)] = self.D.trunk(background)
        finally:
            if stages is not None:
                body.stages = None
            body.injected = None
            for p, w in zip(trunk_params, was):
                p.requires_grad_(w)

    def _cached_trunk(self, background, detach, pop):
        key = self._bg_key(background)
        out = self._trunk_cache.pop(key, None) if pop else self._trunk_cache.get(key)
        if out is None or not detach:
            return out
        from ..detr_util.misc import NestedTensor
        feats, pos = out
        return [NestedTensor(f.tensors.detach(), f.mask, getattr(f, 'uniform', False)) for f in feats], [p.detach() for p in pos]

    def _dual_trunks(self, background):
        """G's and D's trunk on `background` as grouped launches (detr_backbone.dual_trunk_forward), parked on the two bodies for the G and D forwards
        that follow in this call.  Used where a phase evaluates both trunks itself (the reference's call pattern and phase-level sharing: Gmain's
        G + D(fake), Dmain's G + first D pass); each module's graph is built by its own replayed forward, so a frozen module gets none."""
        if core.knob('DUAL_TRUNK', 1) == 0 or not isinstance(background, torch.Tensor) or not hasattr(self.G, 'backbone') or not hasattr(self.D, 'backbone'):
            return
        g_body, d_body = self.G.backbone[0].body, self.D.backbone[0].body
        if type(g_body) is not type(d_body) or not hasattr(g_body, '_entrance') or g_body.stages is not None or d_body.stages is not None:
            return
        from .detr_backbone import dual_trunk_forward
        key = (background.data_ptr(), tuple(background.shape))
        out_g, out_d = dual_trunk_forward(g_body, d_body, background, background)
        g_body.injected = {key: out_g}
        d_body.injected = {key: out_d}

    def _drop_parked_trunks(self):
        for m in (self.G, self.D):
            if hasattr(m, 'backbone') and getattr(m.backbone[0].body, 'injected', None):
                m.backbone[0].body.injected = None

    def run_G(self, z, bbox_class, bbox_real, bbox_text, bbox_patch, padding_mask, background, c, reconst=False, update_emas=False):
        if not reconst:
            return self.G(z, bbox_class, bbox_real, bbox_text, bbox_patch, padding_mask, background, c)
        return self.G(z, bbox_class, bbox_real, bbox_text, bbox_patch, padding_mask, background, c, reconst)

    def run_D(self, bbox, bbox_class, bbox_text, bbox_patch, padding_mask, background, c, reconst=False, blur_sigma=0, update_emas=False,
              trunk_out=None):
        kw = {} if trunk_out is None else dict(trunk_out=trunk_out)
        if not reconst:
            return self.D(bbox, bbox_class, bbox_text, bbox_patch, padding_mask, background, c, **kw)
        return self.D(bbox, bbox_class, bbox_text, bbox_patch, padding_mask, background, c, reconst, **kw)

    def g_main_loss(self, bbox_real, bbox_class, bbox_text, bbox_patch, padding_mask, background, gen_z, gen_c, gain=1.0):
        w = self.w
        valid = ~padding_mask
        static = bool(getattr(self.G, 'static_shapes', False))
        cached = self._cached_trunk(background, detach=True, pop=False)
        if cached is None:
            self._dual_trunks(background)      # G's trunk and the trunk of D(fake) in one pass
        bbox_fake, loss_z, cls_logits, loss_lm, loss_text_len = self.run_G(gen_z, bbox_class, bbox_real, bbox_text, bbox_patch, padding_mask, background, gen_c, reconst=True)
        gen_logits, gen_logits_uncond = self.run_D(bbox_fake, bbox_class, bbox_text, bbox_patch, padding_mask, background, gen_c,
                                                   trunk_out=cached)
        T = hl.Term
        w_ = lambda v, key: v * w[key]      # noqa: E731
        fused_tail = self.fused_loss_tail and gen_logits.is_cuda
        lay = None
        if static and self.fused_layout_losses and bbox_fake.is_cuda and bbox_fake.shape[1] <= 64:
            # one launch for the four layout terms and their gradients (csrc/layout_loss.hip) instead of ~280 elementwise ones
            if fused_tail:
                lay = layout_losses_per_sample(bbox_fake, bbox_real, valid)
            else:
                l_rec, l_giou, l_ovl, l_aln = layout_losses_fused(bbox_fake, bbox_real, valid)
        elif static:
            l_rec, l_giou = _masked_mse(bbox_fake, bbox_real, valid), _masked_giou(bbox_fake, bbox_real, valid)
            l_ovl, l_aln = compute_overlap(bbox_fake, valid), compute_alignment(bbox_fake, valid)
        else:
            l_rec, l_giou = F.mse_loss(bbox_fake[valid], bbox_real[valid]), generalized_iou_loss(bbox_fake[valid], bbox_real[valid])
            l_ovl, l_aln = compute_overlap(bbox_fake, valid), compute_alignment(bbox_fake, valid)
        self.report('Loss/scores/fake', gen_logits)
        if self._reporting:
            self.report('Loss/signs/fake', gen_logits.sign())
        if fused_tail:
            # the whole tail -- softplus of the two scores, the weights, the sum over the terms, the batch mean and the gain -- and its backward as one
            # launch per direction (hip.losses.combine) instead of ~30 + ~40 scalar-sized ATen launches
            terms = [T('loss_Ggen', gen_logits, 1.0, hl.SOFTPLUS_NEG), T('loss_Ggen_uncond', gen_logits_uncond, 1.0, hl.SOFTPLUS_NEG)]
            if lay is not None:
                terms.append(T(['loss_Ggen_bbox_rec', 'loss_Ggen_bbox_gIoU', 'loss_Ggen_overlapping', 'loss_Ggen_alignment'], lay,
                               [w['Ggen_bbox_rec'], w['Ggen_bbox_gIoU'], w['Ggen_overlapping'], w['Ggen_alignment']], hl.IDENT, [True, True, False, False]))
            else:
                terms += [T('loss_Ggen_bbox_rec', l_rec, w['Ggen_bbox_rec']), T('loss_Ggen_bbox_gIoU', l_giou, w['Ggen_bbox_gIoU']),
                          T('loss_Ggen_overlapping', l_ovl, w['Ggen_overlapping']), T('loss_Ggen_alignment', l_aln, w['Ggen_alignment'])]
            terms.append(T('loss_Ggen_z_rec', loss_z, w['Ggen_z_rec']))
            if static:
                terms.append(T('loss_Ggen_bbox_cls', _masked_ce(cls_logits, bbox_class, valid, raw=True), w['Ggen_bbox_cls'], hl.RATIO))
            else:
                terms.append(T('loss_Ggen_bbox_cls', F.cross_entropy(cls_logits, bbox_class[valid]), w['Ggen_bbox_cls']))
            terms += [T('loss_Ggen_text_rec', loss_lm, w['Ggen_text_rec']), T('loss_Ggen_text_len_rec', loss_text_len, w['Ggen_text_len_rec'])]
            total, rep = hl.combine(terms, gain)
            for k, v in rep.items():
                self.report('Loss/G/' + k, v)
            self.last = dict(bbox_fake=bbox_fake.detach(), **{k: v.detach() for k, v in rep.items()})
            return total
        terms = dict(
            loss_Ggen=F.softplus(-gen_logits),
            loss_Ggen_uncond=F.softplus(-gen_logits_uncond),
            loss_Ggen_bbox_rec=w_(l_rec, 'Ggen_bbox_rec'),
            loss_Ggen_bbox_gIoU=w_(l_giou, 'Ggen_bbox_gIoU'),
            loss_Ggen_overlapping=w_(l_ovl, 'Ggen_overlapping'),
            loss_Ggen_alignment=w_(l_aln, 'Ggen_alignment'),
            loss_Ggen_z_rec=w_(loss_z, 'Ggen_z_rec'),
            loss_Ggen_bbox_cls=w_(_masked_ce(cls_logits, bbox_class, valid) if static else F.cross_entropy(cls_logits, bbox_class[valid]), 'Ggen_bbox_cls'),
            loss_Ggen_text_rec=w_(loss_lm, 'Ggen_text_rec'),
            loss_Ggen_text_len_rec=w_(loss_text_len, 'Ggen_text_len_rec'),
        )
        for k, v in terms.items():
            self.report('Loss/G/' + k, v)
        total = sum(terms.values())
        self.last = dict(bbox_fake=bbox_fake.detach(), **{k: v.detach() for k, v in terms.items()})
        return total.mean().mul(gain)

    def d_gen_terms(self, bbox_real, bbox_class, bbox_text, bbox_patch, padding_mask, background, gen_z, gen_c, trunk_out=None, gen_out=None):
        """-> the terms of loss.py:146-160 (D on the generated layout) as hip.losses.Term objects."""
        if gen_out is None:
            bbox_fake = self.run_G(gen_z, bbox_class, bbox_real, bbox_text, bbox_patch, padding_mask, background, gen_c, update_emas=True)
            gen_out = self.run_D(bbox_fake, bbox_class, bbox_text, bbox_patch, padding_mask, background, gen_c, update_emas=True, trunk_out=trunk_out)
        gen_logits, gen_logits_uncond = gen_out
        self.report('Loss/scores/fake', gen_logits)
        if self._reporting:
            self.report('Loss/signs/fake', gen_logits.sign())
        return [hl.Term('loss_Dgen', gen_logits, 1.0, hl.SOFTPLUS), hl.Term('loss_Dgen_uncond', gen_logits_uncond, 1.0, hl.SOFTPLUS)]

    def d_real_terms(self, bbox_real, bbox_class, bbox_text, bbox_patch, padding_mask, background, real_c, trunk_out=None, real_out=None):
        """-> the terms of loss.py:162-218 (D on the real layout with the reconstruction heads)."""
        w = self.w
        valid = ~padding_mask
        static = bool(getattr(self.D, 'static_shapes', False))
        bbox_real_tmp = bbox_real.detach()
        if real_out is None:
            real_out = self.run_D(bbox_real_tmp, bbox_class, bbox_text, bbox_patch, padding_mask, background, real_c, reconst=True, trunk_out=trunk_out)
        (real_logits, real_logits_uncond, bbox_rec, cls_logits, loss_lm, loss_text_len, bg_rec, bbox_rec_uncond, cls_logits_uncond) = real_out
        T = hl.Term
        fused = self.fused_loss_tail and real_logits.is_cuda and static

        def ce(logits):
            if fused:
                return dict(x=_masked_ce(logits, bbox_class, valid, raw=True), fn=hl.RATIO)
            return dict(x=_masked_ce(logits, bbox_class, valid) if static else F.cross_entropy(logits, bbox_class[valid]))
        mse = (lambda a: _masked_mse(a, bbox_real_tmp, valid)) if static else (lambda a: F.mse_loss(a, bbox_real_tmp[valid]))
        self.report('Loss/scores/real', real_logits)
        if self._reporting:
            self.report('Loss/signs/real', real_logits.sign())
        return [T('loss_Dreal', real_logits, 1.0, hl.SOFTPLUS_NEG), T('loss_Dreal_uncond', real_logits_uncond, 1.0, hl.SOFTPLUS_NEG),
                T('loss_Dreal_bbox_rec', mse(bbox_rec), w['Dreal_bbox_rec']), T('loss_Dreal_bbox_cls', weight=w['Dreal_bbox_cls'], **ce(cls_logits)),
                T('loss_Dreal_text_rec', loss_lm, w['Dreal_text_rec']), T('loss_Dreal_text_len_rec', loss_text_len, w['Dreal_text_len_rec']),
                T('loss_Dreal_bg_rec', F.mse_loss(bg_rec, background), w['Dreal_im_rec']),
                T('loss_Dreal_bbox_rec_uncond', mse(bbox_rec_uncond), w['Dreal_bbox_rec']),
                T('loss_Dreal_bbox_cls_uncond', weight=w['Dreal_bbox_cls'], **ce(cls_logits_uncond))]

    def _finish(self, terms, prefix, gain=1.0):
        """sum of the terms -> batch mean -> x gain (loss.py:213, 253 + the .mul(gain) of :116, 160, 218), every term reported like the reference does.
        On the GPU one launch per direction for all of it (hip.losses.combine)."""
        if self.fused_loss_tail and terms[0].x.is_cuda:
            total, rep = hl.combine(terms, gain)
        else:
            f = {hl.IDENT: lambda x: x, hl.SOFTPLUS: F.softplus, hl.SOFTPLUS_NEG: lambda x: F.softplus(-x)}
            rep = {t.name: f[t.fn](t.x) * t.weight for t in terms}
            total = sum(rep.values()).mean().mul(gain)
        for k, v in rep.items():
            self.report(prefix + k, v)
        return total

    def d_gen_loss(self, *args, gain=1.0, **kwargs):
        return self._finish(self.d_gen_terms(*args, **kwargs), 'Loss/D/', gain)

    def d_real_loss(self, *args, gain=1.0, **kwargs):
        return self._finish(self.d_real_terms(*args, **kwargs), 'Loss/D/', gain)

    def accumulate_gradients(self, phase, bbox_real, bbox_class, bbox_text, bbox_patch, padding_mask, background, real_c, gen_z, gen_c, gain, cur_nimg):
        assert phase in ['Gmain', 'Greg', 'Gboth', 'Dmain', 'Dreg', 'Dboth']
        if self.share_D_trunk != 'iteration':
            self._trunk_cache.clear()          # nothing is carried across phases unless the iteration driver parked a trunk for this iteration
        if self.pl_weight == 0:
            phase = {'Greg': 'none', 'Gboth': 'Gmain'}.get(phase, phase)
        if self.r1_gamma == 0:
            phase = {'Dreg': 'none', 'Dboth': 'Dmain'}.get(phase, phase)
        g_body = self.G.backbone[0].body if hasattr(self.G, 'backbone') else None
        if phase not in ('Gmain', 'Gboth') and g_body is not None and getattr(g_body, 'injected', None):
            g_body.injected = None     # a parked G-trunk evaluation belongs to this iteration's Gmain only
        # A trunk evaluation parked for this call (detr_backbone.ResNet50Body.injected, keyed by the batch's address) must not outlive it: after a miss
        # or an exception a later batch at the same address would otherwise be served stale features.  (Iteration-level sharing parks G's trunks of
        # ALL micro-batches before Gmain: those stay until the phase after Gmain begins, see above.)
        if isinstance(background, torch.Tensor) and background.is_cuda:
            core.zero_arena_begin(background.device)      # one fill for the phase's small accumulation targets (hip.core._ZeroArena)
        try:
            self._run_phase(phase, bbox_real, bbox_class, bbox_text, bbox_patch, padding_mask, background, real_c, gen_z, gen_c, gain)
        except BaseException:
            self._drop_parked_trunks()
            raise
        finally:
            core.zero_arena_end()
        if self.share_D_trunk != 'iteration' or phase not in ('Gmain', 'Gboth'):
            self._drop_parked_trunks()

    def g_pl_loss(self, bbox_real, bbox_class, bbox_text, bbox_patch, padding_mask, background, gen_z, gen_c, gain=1.0):
        """Path-length regularisation (loss.py:119-142): the first `batch / pl_batch_shrink` samples, || d(bbox_fake . noise) / d z || against
        its running mean.  G's heads and layout decoder run on hip/composite.py (differentiated twice)."""
        bs = gen_z.shape[0] // self.pl_batch_shrink
        z = gen_z[:bs].detach().requires_grad_(True)
        with composite.higher_order():
            bbox_fake = self.run_G(z, bbox_class[:bs], bbox_real[:bs], bbox_text[:bs], bbox_patch[:bs], padding_mask[:bs], background[:bs], gen_c[:bs])
        noise = self.pl_noise_fn(bbox_fake) if self.pl_noise_fn is not None else torch.randn_like(bbox_fake)
        pl_noise = noise / float(bbox_fake.shape[2])
        pl_grads = torch.autograd.grad(outputs=[(bbox_fake * pl_noise).sum()], inputs=[z], create_graph=True, only_inputs=True)[0]
        pl_lengths = pl_grads.square().sum([1, 2]).sqrt()
        pl_mean = self.pl_mean.lerp(pl_lengths.mean(), self.pl_decay)
        self.pl_mean.copy_(pl_mean.detach())
        pl_penalty = (pl_lengths - pl_mean).square()
        self.report('Loss/pl_penalty', pl_penalty)
        loss_Gpl = pl_penalty * self.pl_weight
        self.report('Loss/G/reg', loss_Gpl)
        self.last = dict(pl_penalty=pl_penalty.detach(), pl_lengths=pl_lengths.detach(), pl_grads=pl_grads.detach())
        return loss_Gpl.mean().mul(gain)

    def d_r1_loss(self, bbox_real, bbox_class, bbox_text, bbox_patch, padding_mask, background, real_c, gain=1.0):
        """R1 (loss.py:162-166, 207-217 in phase 'Dreg'): gamma / 2 * || d D(real) / d bbox_real ||^2 per sample.  The reference evaluates
        D with reconst=True here and discards the seven reconstruction outputs; only the conditional score is formed.  D's `fc_bbox`,
        `enc_fc_in`, layout decoder and `fc_out_disc` run on hip/composite.py (differentiated twice)."""
        bbox_real_tmp = bbox_real.detach().requires_grad_(True)
        with composite.higher_order():
            real_logits, _ = self.run_D(bbox_real_tmp, bbox_class, bbox_text, bbox_patch, padding_mask, background, real_c)
        self.report('Loss/scores/real', real_logits)
        if self._reporting:
            self.report('Loss/signs/real', real_logits.sign())
        r1_grads = torch.autograd.grad(outputs=[real_logits.sum()], inputs=[bbox_real_tmp], create_graph=True, only_inputs=True)[0]
        r1_penalty = r1_grads.square().sum([1, 2])
        loss_Dr1 = r1_penalty * (self.r1_gamma / 2)
        self.report('Loss/r1_penalty', r1_penalty)
        self.report('Loss/D/reg', loss_Dr1)
        self.last = dict(r1_penalty=r1_penalty.detach(), r1_grads=r1_grads.detach(), real_logits=real_logits.detach())
        return loss_Dr1.mean().mul(gain)

    def _run_phase(self, phase, bbox_real, bbox_class, bbox_text, bbox_patch, padding_mask, background, real_c, gen_z, gen_c, gain):
        # 'Gboth' / 'Dboth' (no lazy regularisation: reg_interval None) = the main phase, then the regulariser on a forward pass of its own.
        # The reference shares Dreal's forward with R1 in 'Dboth' (loss.py:162-217): same expected gradient, independent dropout draws here.
        if phase in ('Greg', 'Gboth'):
            if phase == 'Gboth':
                self._run_phase('Gmain', bbox_real, bbox_class, bbox_text, bbox_patch, padding_mask, background, real_c, gen_z, gen_c, gain)
            self.g_pl_loss(bbox_real, bbox_class, bbox_text, bbox_patch, padding_mask, background, gen_z, gen_c, gain=gain).backward()
        if phase in ('Dreg', 'Dboth'):
            if phase == 'Dboth':
                self._run_phase('Dmain', bbox_real, bbox_class, bbox_text, bbox_patch, padding_mask, background, real_c, gen_z, gen_c, gain)
            self.d_r1_loss(bbox_real, bbox_class, bbox_text, bbox_patch, padding_mask, background, real_c, gain=gain).backward()
        if phase == 'Gmain':
            self.g_main_loss(bbox_real, bbox_class, bbox_text, bbox_patch, padding_mask, background, gen_z, gen_c, gain=gain).backward()
        if phase == 'Dmain':
            if self.share_D_trunk and hasattr(self.D, 'trunk'):   # True / 'phase' / 'iteration'
                cached = self._cached_trunk(background, detach=False, pop=True)
                if cached is None:
                    self._dual_trunks(background)      # the generator's (no-grad) trunk beside the phase's one D-trunk evaluation
                if self.pair_D_passes and hasattr(self.D, 'forward_pair'):
                    # both D passes of the phase as ONE batch of 2B layouts (Discriminator.forward_pair): half the transformer / head launches
                    bbox_fake = self.run_G(gen_z, bbox_class, bbox_real, bbox_text, bbox_patch, padding_mask, background, gen_c, update_emas=True)
                    trunk = cached if cached is not None else self.D.trunk(background)
                    gen_out, real_out = self.D.forward_pair(bbox_fake, bbox_real.detach(), bbox_class, bbox_text, bbox_patch, padding_mask, background, real_c, trunk_out=trunk)
                    t_gen = self.d_gen_terms(bbox_real, bbox_class, bbox_text, bbox_patch, padding_mask, background, gen_z, gen_c, gen_out=gen_out)
                    t_real = self.d_real_terms(bbox_real, bbox_class, bbox_text, bbox_patch, padding_mask, background, real_c, real_out=real_out)
                elif cached is not None:
                    t_gen = self.d_gen_terms(bbox_real, bbox_class, bbox_text, bbox_patch, padding_mask, background, gen_z, gen_c, trunk_out=cached)
                    t_real = self.d_real_terms(bbox_real, bbox_class, bbox_text, bbox_patch, padding_mask, background, real_c, trunk_out=cached)
                else:
                    trunk = self.D.trunk(background)
                    t_gen = self.d_gen_terms(bbox_real, bbox_class, bbox_text, bbox_patch, padding_mask, background, gen_z, gen_c, trunk_out=trunk)
                    t_real = self.d_real_terms(bbox_real, bbox_class, bbox_text, bbox_patch, padding_mask, background, real_c, trunk_out=trunk)
                # one backward: the trunk sees the summed gradient of both passes; mean(gen terms) + mean(real terms) = one combine over all of them
                self._finish(t_gen + t_real, 'Loss/D/', gain).backward()
            else:
                self._dual_trunks(background)          # reference call pattern: the generator's trunk beside the trunk of D(fake); D(real) evaluates its own
                self.d_gen_loss(bbox_real, bbox_class, bbox_text, bbox_patch, padding_mask, background, gen_z, gen_c, gain=gain).backward()
                self.d_real_loss(bbox_real, bbox_class, bbox_text, bbox_patch, padding_mask, background, real_c, gain=gain).backward()
