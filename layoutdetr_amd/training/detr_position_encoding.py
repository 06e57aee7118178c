"""Sine position embedding (reference: training/detr_position_encoding.py:22-58).

The embedding depends only on the padding mask.  For uniform-size batches (mask all False — every
training batch: detr_util/misc.py:336-339) it is a constant of (B, h, w), so it is computed once and
cached instead of re-running ~12 elementwise kernels per forward."""
import math

import torch
from torch import nn

from ..detr_util.misc import NestedTensor


class PositionEmbeddingSine(nn.Module):
    def __init__(self, num_pos_feats=64, temperature=10000, normalize=False, scale=None):
        super().__init__()
        self.num_pos_feats = num_pos_feats
        self.temperature = temperature
        self.normalize = normalize
        if scale is not None and normalize is False:
            raise ValueError('normalize should be True if scale is passed')
        self.scale = 2 * math.pi if scale is None else scale
        self._cache = {}

    def _compute(self, mask, device):
        not_mask = ~mask
        y_embed = not_mask.cumsum(1, dtype=torch.float32)
        x_embed = not_mask.cumsum(2, dtype=torch.float32)
        if self.normalize:
            eps = 1e-6
            y_embed = y_embed / (y_embed[:, -1:, :] + eps) * self.scale
            x_embed = x_embed / (x_embed[:, :, -1:] + eps) * self.scale
        dim_t = torch.arange(self.num_pos_feats, dtype=torch.float32, device=device)
        dim_t = self.temperature ** (2 * torch.div(dim_t, 2, rounding_mode='floor') / self.num_pos_feats)
        pos_x = x_embed[:, :, :, None] / dim_t
        pos_y = y_embed[:, :, :, None] / dim_t
        pos_x = torch.stack((pos_x[:, :, :, 0::2].sin(), pos_x[:, :, :, 1::2].cos()), dim=4).flatten(3)
        pos_y = torch.stack((pos_y[:, :, :, 0::2].sin(), pos_y[:, :, :, 1::2].cos()), dim=4).flatten(3)
        # [B, h, w, 2F] is already the NHWC row layout the transformer consumes; expose the NCHW view.
        return torch.cat((pos_y, pos_x), dim=3).contiguous().permute(0, 3, 1, 2)

    def forward(self, tensor_list: NestedTensor):
        mask = tensor_list.mask
        assert mask is not None
        if getattr(tensor_list, 'uniform', False):
            key = (tuple(mask.shape), str(mask.device))
            if key not in self._cache:
                self._cache[key] = self._compute(mask, mask.device)
            return self._cache[key]
        return self._compute(mask, mask.device)

    def __deepcopy__(self, memo):
        new = PositionEmbeddingSine(self.num_pos_feats, self.temperature, self.normalize, self.scale)
        new.training = self.training
        return new

    def __getstate__(self):
        d = dict(self.__dict__)
        d['_cache'] = {}
        return d
