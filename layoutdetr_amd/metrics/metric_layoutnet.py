"""Layout losses on the generator path (reference: metrics/metric_layoutnet.py compute_overlap :153-179,
compute_alignment :182-201, generalized_iou_loss :245-275; util.convert_xywh_to_ltrb util.py:62-68).
Same names, arguments and values; written without in-place masked writes so they stay autograd-safe.
These are tiny [B, 9, 9] tensor programs (device-side glue around the hot kernels)."""
import torch


def convert_xywh_to_ltrb(bbox):
    xc, yc, w, h = bbox
    return [xc - w / 2, yc - h / 2, xc + w / 2, yc + h / 2]


def compute_overlap(bbox, mask):
    bbox = bbox.masked_fill(~mask.unsqueeze(-1), 0).permute(2, 0, 1)
    l1, t1, r1, b1 = convert_xywh_to_ltrb(bbox.unsqueeze(-1))
    l2, t2, r2, b2 = convert_xywh_to_ltrb(bbox.unsqueeze(-2))
    a1 = (r1 - l1) * (b1 - t1)
    l_max, r_min = torch.maximum(l1, l2), torch.minimum(r1, r2)
    t_max, b_min = torch.maximum(t1, t2), torch.minimum(b1, b2)
    cond = (l_max < r_min) & (t_max < b_min)
    ai = torch.where(cond, (r_min - l_max) * (b_min - t_max), torch.zeros_like(a1[0]))
    diag_mask = torch.eye(a1.size(1), dtype=torch.bool, device=a1.device)
    ai = ai.masked_fill(diag_mask, 0)
    ar = torch.nan_to_num(ai / a1)
    return ar.sum(dim=(1, 2)) / mask.float().sum(-1)


def compute_alignment(bbox, mask):
    bbox = bbox.permute(2, 0, 1)
    xl, yt, xr, yb = convert_xywh_to_ltrb(bbox)
    xc, yc = bbox[0], bbox[1]
    X = torch.stack([xl, xc, xr, yt, yc, yb], dim=1)
    X = X.unsqueeze(-1) - X.unsqueeze(-2)
    n = X.size(2)
    eye = torch.eye(n, dtype=torch.bool, device=X.device)
    X = torch.where(eye, torch.ones_like(X), X).abs().permute(0, 2, 1, 3)
    X = torch.where(mask[:, :, None, None], X, torch.ones_like(X))
    X = X.min(-1).values.min(-1).values
    X = torch.where(X.eq(1.), torch.zeros_like(X), X)
    X = -torch.log(1 - X)
    return X.sum(-1) / mask.float().sum(-1)


def generalized_iou_loss(layout_1, layout_2):
    l1, t1, r1, b1 = convert_xywh_to_ltrb(layout_1.T)
    l2, t2, r2, b2 = convert_xywh_to_ltrb(layout_2.T)
    a1, a2 = (r1 - l1) * (b1 - t1), (r2 - l2) * (b2 - t2)
    l_max, r_min = torch.maximum(l1, l2), torch.minimum(r1, r2)
    t_max, b_min = torch.maximum(t1, t2), torch.minimum(b1, b2)
    cond = (l_max < r_min) & (t_max < b_min)
    ai = torch.where(cond, (r_min - l_max) * (b_min - t_max), torch.zeros_like(a1))
    au = a1 + a2 - ai
    iou = ai / au
    l_min, r_max = torch.minimum(l1, l2), torch.maximum(r1, r2)
    t_min, b_max = torch.minimum(t1, t2), torch.maximum(b1, b2)
    ah = (r_max - l_min) * (b_max - t_min)
    g_iou = iou - (ah - au) / ah
    return (1.0 - g_iou).mean()


class _LayoutLossesFn(torch.autograd.Function):
    """The four layout losses of the generator phase in one launch (csrc/layout_loss.hip): values and d/d bbox together; the
    backward is one weighted sum of the saved gradients.  Returns per-sample shares [4, B]."""

    @staticmethod
    def forward(ctx, bbox, bbox_ref, valid):
        from ..hip import core
        core.require_gpu(bbox, bbox_ref, valid)
        B, N, _ = bbox.shape
        x = bbox.detach().to(torch.float32).contiguous()
        r = bbox_ref.detach().to(torch.float32).contiguous()
        v = valid.contiguous().view(torch.uint8) if valid.dtype == torch.bool else valid.to(torch.uint8).contiguous()
        losses = torch.empty((4, B), device=x.device, dtype=torch.float32)
        grads = torch.empty((4, B, N, 4), device=x.device, dtype=torch.float32)
        core.check(core.lib().ldetr_layout_losses_f32(core.ptr(x), core.ptr(r), core.ptr(v), B, N, core.ptr(losses), core.ptr(grads),
                                                      core.stream()), 'layout_losses')
        ctx.save_for_backward(grads)
        ctx.shape = (B, N)
        return losses

    @staticmethod
    def backward(ctx, g):
        from ..hip import core
        grads, = ctx.saved_tensors
        B, N = ctx.shape
        g = g.to(torch.float32).contiguous()
        d = torch.empty((B, N, 4), device=g.device, dtype=torch.float32)
        core.check(core.lib().ldetr_layout_losses_bwd_f32(core.ptr(grads), core.ptr(g), B, N, core.ptr(d), core.stream()), 'layout_losses_bwd')
        return d, None, None


def layout_losses_fused(bbox_fake, bbox_real, valid):
    """(mse_loss(fake[valid], real[valid]), generalized_iou_loss(fake[valid], real[valid]), compute_overlap(fake, valid) [B],
    compute_alignment(fake, valid) [B]) with gradients to bbox_fake only; N <= 64 boxes per sample, GPU tensors."""
    if bbox_real.requires_grad:
        raise NotImplementedError('layout_losses_fused: the reference boxes are data (no gradient is produced for them)')
    out = _LayoutLossesFn.apply(bbox_fake, bbox_real, valid)
    return out[0].sum(), out[1].sum(), out[2], out[3]


def layout_losses_per_sample(bbox_fake, bbox_real, valid):
    """-> [4, B]: per-sample shares of (mse, gIoU) -- their sums are the two scalar terms -- and the per-sample overlap / alignment terms, as ONE
    autograd tensor (hip.losses.combine takes it whole)."""
    if bbox_real.requires_grad:
        raise NotImplementedError('layout_losses_fused: the reference boxes are data (no gradient is produced for them)')
    return _LayoutLossesFn.apply(bbox_fake, bbox_real, valid)


def linear_sum_assignment_batched(cost, maximize=False):
    """Batched Hungarian on device: cost [batch, n, n] float64 -> (row_ind, col_ind) int32 [batch, n], bit-exact with
    scipy.optimize.linear_sum_assignment as used by compute_maximum_iou_for_layout (metric_layoutnet.py:100-113)."""
    from ..hip import core
    core.require_gpu(cost)
    c = cost.to(torch.float64).contiguous()
    batch, n, _ = c.shape
    ri = torch.empty((batch, n), dtype=torch.int32, device=c.device)
    ci = torch.empty_like(ri)
    core.check(core.lib().ldetr_lsap_f64(core.ptr(c), batch, n, 1 if maximize else 0, core.ptr(ri), core.ptr(ci), core.stream()), 'lsap')
    return ri, ci


# ------------------------------------------------------------------------------------------------------------------------
# Evaluation metrics (reference metrics/metric_layoutnet.py:66-150, 204-242).  Same names and arguments as the
# reference.  The reference scores one pair of layouts at a time on the host (numpy + scipy, a process pool over conditions);
# here every same-condition pair of a corpus is scored in ONE batched device pass: pairwise IoU / DocSim weights as [P, n, n]
# tensors, cross-label entries excluded by a large cost, one device Hungarian solve per pair (csrc/lsap.hip, n <= 64).
# Since both layouts of a pair carry the same multiset of labels, the optimum of the single n x n problem is the sum of the
# reference's per-label optima.  Only the final N x M matching per condition (rectangular, N and M unbounded) stays on scipy.
_CROSS_LABEL = -1.0e6


def compute_iou(box_1, box_2):
    """IoU of corresponding xywh boxes: [N, 4] x [N, 4] -> [N] (reference :66-92); torch tensors on any device or numpy."""
    import numpy as np
    if isinstance(box_1, np.ndarray):
        return compute_iou(torch.from_numpy(box_1), torch.from_numpy(box_2)).numpy()
    l1, t1, r1, b1 = convert_xywh_to_ltrb(box_1.T)
    l2, t2, r2, b2 = convert_xywh_to_ltrb(box_2.T)
    a1, a2 = (r1 - l1) * (b1 - t1), (r2 - l2) * (b2 - t2)
    l_max, r_min = torch.maximum(l1, l2), torch.minimum(r1, r2)
    t_max, b_min = torch.maximum(t1, t2), torch.minimum(b1, b2)
    cond = (l_max < r_min) & (t_max < b_min)
    ai = torch.where(cond, (r_min - l_max) * (b_min - t_max), torch.zeros_like(a1))
    return torch.nan_to_num(ai / (a1 + a2 - ai))


def compute_docsim_weight(box_1, box_2):
    """DocSim weight of corresponding boxes (reference :204-221)."""
    import numpy as np
    if isinstance(box_1, np.ndarray):
        return compute_docsim_weight(torch.from_numpy(box_1), torch.from_numpy(box_2)).numpy()
    xc1, yc1, w1, h1 = box_1.T
    xc2, yc2, w2, h2 = box_2.T
    location_difference = ((xc1 - xc2) ** 2 + (yc1 - yc2) ** 2) ** 0.5
    shape_difference = (w1 - w2).abs() + (h1 - h2).abs()
    area_factor = torch.minimum(w1 * h1, w2 * h2) ** 0.5
    return area_factor * 2 ** (-location_difference - 2.0 * shape_difference)


def compute_iou_for_layout(layout_1, layout_2):
    (bi, li), (bj, lj) = layout_1, layout_2
    return compute_iou(bi, bj).mean().item()


def compute_docsim_for_layout(layout_1, layout_2):
    (bi, li), (bj, lj) = layout_1, layout_2
    return compute_docsim_weight(bi, bj).mean().item()


def maximum_scores_batched(b1, l1, b2, l2, kind='iou'):
    """Scores of P layout pairs at once.  b1, b2: [P, n, 4] float32 device tensors (xywh); l1, l2: [P, n] integer labels with
    equal label multisets per pair; n <= 64.  -> [P] float64: per pair, the maximum over label-preserving matchings of the
    summed IoU (kind='iou', reference :100-113) or DocSim weight (kind='docsim', :229-242), divided by n."""
    P, n, _ = b1.shape
    fn = compute_iou if kind == 'iou' else compute_docsim_weight
    # rows = boxes of layout 2, columns = boxes of layout 1 (the reference's meshgrid order)
    bi = b1[:, None, :, :].expand(P, n, n, 4).reshape(-1, 4)
    bj = b2[:, :, None, :].expand(P, n, n, 4).reshape(-1, 4)
    w = fn(bi, bj).reshape(P, n, n).to(torch.float64)
    same = l2[:, :, None] == l1[:, None, :]
    cost = torch.where(same, w, torch.full_like(w, _CROSS_LABEL))
    ri, ci = linear_sum_assignment_batched(cost, maximize=True)
    picked = torch.gather(cost, 2, ci.long().unsqueeze(-1)).squeeze(-1)        # row_ind is 0..n-1 in order
    if bool((picked <= _CROSS_LABEL / 2).any()):
        raise ValueError('maximum_scores_batched: a pair of layouts does not share one multiset of labels')
    return picked.sum(-1) / n


def _pair_on_device(layout_1, layout_2, kind):
    import numpy as np
    (bi, li), (bj, lj) = layout_1, layout_2
    dev = torch.device('cuda', torch.cuda.current_device())
    t = lambda a, dt: torch.as_tensor(np.asarray(a), dtype=dt, device=dev).unsqueeze(0)
    return maximum_scores_batched(t(bi, torch.float32), t(li, torch.int64), t(bj, torch.float32), t(lj, torch.int64), kind).item()


def compute_maximum_iou_for_layout(layout_1, layout_2):
    """layout = (bbox [N, 4] array, label [N] array), as in the reference (:100-113)."""
    return _pair_on_device(layout_1, layout_2, 'iou')


def compute_maximum_docsim_for_layout(layout_1, layout_2):
    return _pair_on_device(layout_1, layout_2, 'docsim')


def compute_maximum_iou(layouts_1, layouts_2, n_jobs=None):
    """Corpus-level maximum IoU (reference :116-150).  Layouts are grouped by their sorted label list; every same-condition
    pair is scored in one batched device pass per condition; the N x M matching per condition runs on scipy as in the
    reference.  n_jobs is accepted for signature compatibility (there is no process pool: the device pass replaces it).
    Like the reference (:118-124) the pair scores are enumerated layouts_2-major and re-wrapped to (N, M)."""
    import numpy as np
    from scipy.optimize import linear_sum_assignment
    dev = torch.device('cuda', torch.cuda.current_device())

    def group(layout_list):
        out = {}
        for bs, ls in layout_list:
            out.setdefault(str(sorted(np.asarray(ls).tolist())), []).append((np.asarray(bs), np.asarray(ls)))
        return out

    c1, c2 = group(layouts_1), group(layouts_2)
    scores = []
    for key in c1:
        if key not in c2:
            continue
        A, B = c1[key], c2[key]
        N, M = len(A), len(B)
        b1 = torch.as_tensor(np.stack([a[0] for a in A]), dtype=torch.float32, device=dev)
        l1 = torch.as_tensor(np.stack([a[1] for a in A]), dtype=torch.int64, device=dev)
        b2 = torch.as_tensor(np.stack([b[0] for b in B]), dtype=torch.float32, device=dev)
        l2 = torch.as_tensor(np.stack([b[1] for b in B]), dtype=torch.int64, device=dev)
        jj, ii = torch.meshgrid(torch.arange(M, device=dev), torch.arange(N, device=dev), indexing='ij')   # j outer, i inner
        ii, jj = ii.reshape(-1), jj.reshape(-1)
        flat = maximum_scores_batched(b1[ii], l1[ii], b2[jj], l2[jj], 'iou')
        s = flat.cpu().numpy().reshape(N, M)
        r, c = linear_sum_assignment(s, maximize=True)
        scores.extend(s[r, c].tolist())
    return float(np.mean(scores))
