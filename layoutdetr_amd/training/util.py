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


def encode_seq_first_pair(enc_a: TransformerEncoder, xa, enc_b: TransformerEncoder, xb, key_padding_mask):
    """encode_seq_first for two INDEPENDENT stacks of the same geometry (D's conditional and unconditional reconstruction decoders,
    networks_detr.py:269, 275-276 run them one after the other): one launch per sub-block step for both (hip.stacks)."""
    from ..hip import stacks as hstacks
    L, B, E = xa.shape
    if xb.shape == xa.shape:
        pa = enc_a.as_prog(xa.permute(1, 0, 2).reshape(B * L, E), B, L, key_padding_mask)
        pb = enc_b.as_prog(xb.permute(1, 0, 2).reshape(B * L, E), B, L, key_padding_mask)
        if pa is not None and pb is not None:
            ya, yb = hstacks.run([pa, pb])
            return ya.reshape(B, L, E).permute(1, 0, 2), yb.reshape(B, L, E).permute(1, 0, 2)
    return encode_seq_first(enc_a, xa, key_padding_mask), encode_seq_first(enc_b, xb, key_padding_mask)


class TransformerWithToken_layoutganpp(nn.Module):
    def __init__(self, d_model, nhead, dim_feedforward, num_layers):
        super().__init__()
        self.token = nn.Parameter(torch.randn(1, 1, d_model))
        self.register_buffer('token_mask', torch.zeros(1, 1, dtype=torch.bool))
        self.core = TransformerEncoder(TransformerEncoderLayer(d_model=d_model, nhead=nhead, dim_feedforward=dim_feedforward),
                                       num_layers=num_layers)

    def _with_token(self, x, src_key_padding_mask):
        B = x.size(1)
        x = torch.cat([self.token.expand(-1, B, -1), x], dim=0)
        return x, torch.cat([self.token_mask.expand(B, -1), src_key_padding_mask], dim=1)

    def forward(self, x, src_key_padding_mask):
        x, padding_mask = self._with_token(x, src_key_padding_mask)
        return encode_seq_first(self.core, x, padding_mask)

    def as_prog(self, x, src_key_padding_mask):
        """This forward as a hip.stacks.Prog (to run in lock-step with an independent stack: D's layout decoder, networks_detr.py:242-243) plus the
        function that turns the stack's [B*L, E] output into forward()'s [L, B, E]; None when the stack node does not take it."""
        x, padding_mask = self._with_token(x, src_key_padding_mask)
        L, B, E = x.shape
        prog = self.core.as_prog(x.permute(1, 0, 2).reshape(B * L, E), B, L, padding_mask)
        if prog is None:
            return None
        return prog, (lambda y2: y2.reshape(B, L, E).permute(1, 0, 2))
