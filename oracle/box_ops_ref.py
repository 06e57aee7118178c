"""ORACLE (test infrastructure only — never imported by the product path).

CPU restatement (numpy, fp32, one IEEE operation per reference operation) of the reference's detr_util/box_ops.py:
box_cxcywh_to_xyxy (:19-23), box_xyxy_to_cxcywh (:26-30), box_iou (:35-48; torchvision's box_area = (x1-x0)*(y1-y0)),
generalized_box_iou (:51-71), and the DETR matcher's use of it: scipy.optimize.linear_sum_assignment on cost = -GIoU (oracle/lsap.c is
the C restatement of that solver).  Pinned bit-exactly to tests/golden/box_ops.npz (oracle/gen_golden.py:gen_box_ops drives the
imported reference functions)."""
import numpy as np

f32 = np.float32


def box_cxcywh_to_xyxy(x):
    x = np.asarray(x, f32)
    xc, yc, w, h = x[..., 0], x[..., 1], x[..., 2], x[..., 3]
    return np.stack([xc - f32(0.5) * w, yc - f32(0.5) * h, xc + f32(0.5) * w, yc + f32(0.5) * h], -1)


def box_xyxy_to_cxcywh(x):
    x = np.asarray(x, f32)
    x0, y0, x1, y1 = x[..., 0], x[..., 1], x[..., 2], x[..., 3]
    return np.stack([(x0 + x1) / f32(2), (y0 + y1) / f32(2), x1 - x0, y1 - y0], -1)


def box_area(b):
    return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])


def box_iou(b1, b2):
    b1, b2 = np.asarray(b1, f32), np.asarray(b2, f32)
    a1, a2 = box_area(b1), box_area(b2)
    lt = np.maximum(b1[:, None, :2], b2[None, :, :2]); rb = np.minimum(b1[:, None, 2:], b2[None, :, 2:])
    wh = np.clip(rb - lt, 0, None)
    inter = wh[..., 0] * wh[..., 1]
    union = a1[:, None] + a2[None] - inter
    with np.errstate(divide='ignore', invalid='ignore'):
        return inter / union, union


def generalized_box_iou(b1, b2):
    b1, b2 = np.asarray(b1, f32), np.asarray(b2, f32)
    assert (b1[:, 2:] >= b1[:, :2]).all() and (b2[:, 2:] >= b2[:, :2]).all()
    iou, union = box_iou(b1, b2)
    lt = np.minimum(b1[:, None, :2], b2[None, :, :2]); rb = np.maximum(b1[:, None, 2:], b2[None, :, 2:])
    wh = np.clip(rb - lt, 0, None)
    area = wh[..., 0] * wh[..., 1]
    with np.errstate(divide='ignore', invalid='ignore'):
        return iou - (area - union) / area


def hungarian_match_giou(pred_cxcywh, tgt_cxcywh, lsap):
    """lsap(cost float64 [n, n]) -> (rows, cols); cost = -GIoU (fp32 values widened to float64, as the matcher hands them to scipy)."""
    g = generalized_box_iou(box_cxcywh_to_xyxy(pred_cxcywh), box_cxcywh_to_xyxy(tgt_cxcywh))
    return lsap((-g).astype(np.float64)), g
