"""Box utilities of the DETR matcher / criterion on the GPU — same names and semantics as the reference's
`detr_util/box_ops.py` (box_cxcywh_to_xyxy :19-23, box_xyxy_to_cxcywh :26-30, box_iou :35-48, generalized_box_iou :51-71).
north_star names this file ("bbox Hungarian-matched regression head"); LayoutDETR's training step itself never calls it
(SURVEY §0.2), so these are the inference / evaluation-side entry points.

The pairwise matrices come from ONE launch (`ldetr_box_giou_pairwise_f32`, csrc/box_ops.hip): fp32 in the reference's operation
order, bit-identical to the reference's CPU values.  `hungarian_match_giou` chains it into the device LSAP solver
(`ldetr_lsap_f64`, bit-exact with scipy): cost = -GIoU in float64 written by the same launch, assignment indices out, no host trip.
"""
import torch

from ..hip import core


def box_cxcywh_to_xyxy(x):
    x_c, y_c, w, h = x.unbind(-1)
    return torch.stack([x_c - 0.5 * w, y_c - 0.5 * h, x_c + 0.5 * w, y_c + 0.5 * h], dim=-1)


def box_xyxy_to_cxcywh(x):
    x0, y0, x1, y1 = x.unbind(-1)
    return torch.stack([(x0 + x1) / 2, (y0 + y1) / 2, x1 - x0, y1 - y0], dim=-1)


def _pairwise(boxes1, boxes2, cxcywh, want, cost_sign=None):
    core.require_gpu(boxes1, boxes2)
    batched = boxes1.dim() == 3
    b1 = core.f32c(boxes1 if batched else boxes1[None])
    b2 = core.f32c(boxes2 if batched else boxes2[None])
    if b1.shape[-1] != 4 or b2.shape[-1] != 4 or b1.shape[0] != b2.shape[0]:
        raise ValueError('boxes must be [N, 4] / [M, 4] or [B, N, 4] / [B, M, 4]')
    B, N, M = b1.shape[0], b1.shape[1], b2.shape[1]
    outs = {k: torch.empty((B, N, M), device=b1.device, dtype=torch.float32) for k in want}
    cost = torch.empty((B, N, M), device=b1.device, dtype=torch.float64) if cost_sign is not None else None
    core.check(core.lib().ldetr_box_giou_pairwise_f32(core.ptr(b1), core.ptr(b2), B, N, M, 1 if cxcywh else 0, core.ptr(outs.get('iou')),
                                                      core.ptr(outs.get('uni')), core.ptr(outs.get('giou')), core.ptr(cost),
                                                      float(cost_sign or 0.0), core.stream()), 'box_giou_pairwise')
    outs = {k: (v if batched else v[0]) for k, v in outs.items()}
    return outs, cost


def box_iou(boxes1, boxes2):
    """-> (iou [N, M], union [N, M]); xyxy boxes (a leading batch dimension on both is accepted)."""
    o, _ = _pairwise(boxes1, boxes2, False, ('iou', 'uni'))
    return o['iou'], o['uni']


def generalized_box_iou(boxes1, boxes2):
    """Generalised IoU, [N, M] pairwise; boxes in [x0, y0, x1, y1] format (degenerate boxes are rejected like the reference does)."""
    assert (boxes1[..., 2:] >= boxes1[..., :2]).all()
    assert (boxes2[..., 2:] >= boxes2[..., :2]).all()
    o, _ = _pairwise(boxes1, boxes2, False, ('giou',))
    return o['giou']


def hungarian_match_giou(pred_cxcywh, target_cxcywh, maximize=False):
    """Assignment of predicted to target boxes ([B, n, 4] each, cx cy w h, n <= 64) under cost = -GIoU (the DETR matcher's cost_giou term):
    box_cxcywh_to_xyxy + generalized_box_iou + the float64 cost matrix in one launch, then the batched device Hungarian solve.
    -> (row_ind, col_ind) int32 [B, n] with scipy.optimize.linear_sum_assignment's ordering, and the GIoU matrix [B, n, n]."""
    if pred_cxcywh.dim() != 3 or pred_cxcywh.shape != target_cxcywh.shape:
        raise ValueError('hungarian_match_giou: [B, n, 4] predictions and targets of the same shape')
    B, n = pred_cxcywh.shape[:2]
    o, cost = _pairwise(pred_cxcywh, target_cxcywh, True, ('giou',), cost_sign=-1.0)
    ri = torch.empty((B, n), dtype=torch.int32, device=cost.device); ci = torch.empty_like(ri)
    core.check(core.lib().ldetr_lsap_f64(core.ptr(cost), B, n, 1 if maximize else 0, core.ptr(ri), core.ptr(ci), core.stream()), 'lsap')
    return ri, ci, o['giou']
