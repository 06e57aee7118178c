"""NestedTensor plumbing (reference: detr_util/misc.py:297-342).  Only what the hot path touches."""
from typing import List, Optional

import torch
from torch import Tensor


class NestedTensor(object):
    def __init__(self, tensors, mask: Optional[Tensor], uniform: bool = False):
        self.tensors = tensors
        self.mask = mask
        self.uniform = uniform  # True: every image fills the batch canvas, i.e. mask is all False

    def to(self, device):
        return NestedTensor(self.tensors.to(device), None if self.mask is None else self.mask.to(device), self.uniform)

    def decompose(self):
        return self.tensors, self.mask

    def __repr__(self):
        return str(self.tensors)


def nested_tensor_from_tensor_list(tensor_list):
    """[B,3,H,W] tensor or list of [3,Hi,Wi] tensors -> zero-padded batch + padding mask (True = padding)."""
    if isinstance(tensor_list, torch.Tensor):
        if tensor_list.ndim != 4:
            raise ValueError('not supported')
        b, c, h, w = tensor_list.shape
        mask = torch.zeros((b, h, w), dtype=torch.bool, device=tensor_list.device)
        return NestedTensor(tensor_list, mask, uniform=True)
    if tensor_list[0].ndim != 3:
        raise ValueError('not supported')
    sizes = [list(img.shape) for img in tensor_list]
    c, h, w = [max(s[i] for s in sizes) for i in range(3)]
    b = len(tensor_list)
    tensor = torch.zeros((b, c, h, w), dtype=tensor_list[0].dtype, device=tensor_list[0].device)
    mask = torch.ones((b, h, w), dtype=torch.bool, device=tensor_list[0].device)
    for img, pad_img, m in zip(tensor_list, tensor, mask):
        pad_img[: img.shape[0], : img.shape[1], : img.shape[2]].copy_(img)
        m[: img.shape[1], : img.shape[2]] = False
    uniform = all(s == sizes[0] for s in sizes)
    return NestedTensor(tensor, mask, uniform=uniform)
