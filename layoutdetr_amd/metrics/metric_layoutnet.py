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
