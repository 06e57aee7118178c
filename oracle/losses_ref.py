"""ORACLE (test infrastructure only — never imported by the product path).

CPU fp32 restatement of the layout losses on the generator path and of the DP gradient post-processing:
  convert_xywh_to_ltrb      util.py:62-68
  compute_overlap           metrics/metric_layoutnet.py:153-179
  compute_alignment         metrics/metric_layoutnet.py:182-201
  generalized_iou_loss      metrics/metric_layoutnet.py:245-275
  dp_postprocess            training/training_loop.py:303-312 (flat SUM all-reduce, /W, nan_to_num(0, 1e5, -1e5))
Pinned by tests/test_oracle_golden.py against tests/golden/{losses,dp_step}.npz.
"""
import torch


def xywh_to_ltrb(b):
    xc, yc, w, h = b
    return xc - w / 2, yc - h / 2, xc + w / 2, yc + h / 2


def compute_overlap(bbox, mask):
    """bbox [B,N,4] (xc,yc,w,h), mask [B,N] True = valid -> [B]: mean over valid of sum_j inter(i,j)/area(i), i != j."""
    bbox = bbox.masked_fill(~mask.unsqueeze(-1), 0).permute(2, 0, 1)
    l1, t1, r1, b1 = xywh_to_ltrb(bbox.unsqueeze(-1))
    l2, t2, r2, b2 = xywh_to_ltrb(bbox.unsqueeze(-2))
    a1 = (r1 - l1) * (b1 - t1)
    lm, rm = torch.maximum(l1, l2), torch.minimum(r1, r2)
    tm, bm = torch.maximum(t1, t2), torch.minimum(b1, b2)
    inter = torch.where((lm < rm) & (tm < bm), (rm - lm) * (bm - tm), torch.zeros_like(a1[0]))
    eye = torch.eye(a1.size(1), dtype=torch.bool)
    inter = inter.masked_fill(eye, 0)
    ratio = torch.nan_to_num(inter / a1)
    return ratio.sum(dim=(1, 2)) / mask.float().sum(-1)


def compute_alignment(bbox, mask):
    """-log(1 - min over the 6 edge/centre coordinates and over other boxes of |delta|), 0 when that min is 1."""
    bbox = bbox.permute(2, 0, 1)
    xl, yt, xr, yb = xywh_to_ltrb(bbox)
    X = torch.stack([xl, bbox[0], xr, yt, bbox[1], yb], dim=1)          # [B,6,N]
    D = X.unsqueeze(-1) - X.unsqueeze(-2)                                # [B,6,N,N]
    n = X.size(2)
    eye = torch.eye(n, dtype=torch.bool)
    D = torch.where(eye, torch.ones_like(D), D).abs().permute(0, 2, 1, 3)  # [B,N,6,N]
    D = torch.where(mask[:, :, None, None], D, torch.ones_like(D))
    m = D.min(-1).values.min(-1).values
    m = torch.where(m == 1.0, torch.zeros_like(m), m)
    return (-torch.log(1 - m)).sum(-1) / mask.float().sum(-1)


def generalized_iou_loss(a, b):
    """a, b [M,4] (xc,yc,w,h) -> mean(1 - gIoU)."""
    l1, t1, r1, b1 = xywh_to_ltrb(a.T)
    l2, t2, r2, b2 = xywh_to_ltrb(b.T)
    a1, a2 = (r1 - l1) * (b1 - t1), (r2 - l2) * (b2 - t2)
    lm, rm = torch.maximum(l1, l2), torch.minimum(r1, r2)
    tm, bm = torch.maximum(t1, t2), torch.minimum(b1, b2)
    ai = torch.where((lm < rm) & (tm < bm), (rm - lm) * (bm - tm), torch.zeros_like(a1))
    au = a1 + a2 - ai
    ah = (torch.maximum(r1, r2) - torch.minimum(l1, l2)) * (torch.maximum(b1, b2) - torch.minimum(t1, t2))
    return (1.0 - (ai / au - (ah - au) / ah)).mean()


def dp_postprocess(flat_sum, world):
    """What every rank holds after the exchange step: sum over ranks / world, then nan_to_num."""
    g = flat_sum / world if world > 1 else flat_sum.clone()
    return torch.nan_to_num(g, nan=0.0, posinf=1e5, neginf=-1e5)
