"""LayoutGAN++-style token transformer used by D's unconditional branch (reference: training/util.py:13-43).
The reference wraps nn.TransformerEncoder; here the same parameters drive the HIP encoder layers."""
import torch
import torch.nn as nn

from .detr_transformer import TransformerEncoder, TransformerEncoderLayer


def encode_seq_first(encoder: TransformerEncoder, x, key_padding_mask):
    """x: [L, B, E] (reference layout) -> [L, B, E] through batch-first HIP encoder layers."""
    L, B, E = x.shape
    x2 = x.permute(1, 0, 2).reshape(B * L, E)
    y2 = encoder.forward2d(x2, B, L, key_padding_mask, None)
    return y2.reshape(B, L, E).permute(1, 0, 2)


class TransformerWithToken_layoutganpp(nn.Module):
    def __init__(self, d_model, nhead, dim_feedforward, num_layers):
        super().__init__()
        self.token = nn.Parameter(torch.randn(1, 1, d_model))
        self.register_buffer('token_mask', torch.zeros(1, 1, dtype=torch.bool))
        self.core = TransformerEncoder(TransformerEncoderLayer(d_model=d_model, nhead=nhead, dim_feedforward=dim_feedforward),
                                       num_layers=num_layers)

    def forward(self, x, src_key_padding_mask):
        B = x.size(1)
        x = torch.cat([self.token.expand(-1, B, -1), x], dim=0)
        padding_mask = torch.cat([self.token_mask.expand(B, -1), src_key_padding_mask], dim=1)
        return encode_seq_first(self.core, x, padding_mask)
