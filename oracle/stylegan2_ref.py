"""ORACLE (test infrastructure only — never imported by the product path).

CPU fp32 restatement of the StyleGAN2 `Decoder` chain used as D's background-reconstruction head
(training/networks_stylegan2.py: FullyConnectedLayer :92-126, SynthesisLayer :272-331, ToRGBLayer :336-356,
SynthesisBlock :361-460 ('skip' architecture), SynthesisNetwork :465-520, DecoderMappingNetwork :903-967,
Decoder :972-994), as pure functions over a state dict.  Configuration fixed to what networks_detr.py:261
constructs: use_noise=False, num_fp16_res=0, conv_clamp=None, fused_modconv_default=False.
Pinned by tests/test_oracle_golden.py against tests/golden/decoder.npz.
"""
import math

import torch

from . import ops_ref


def fully_connected(sd, pre, x, activation='linear', lr_multiplier=1.0):
    w = sd[pre + 'weight']
    w = w * (lr_multiplier / math.sqrt(w.shape[1]))
    b = sd.get(pre + 'bias')
    if b is not None and lr_multiplier != 1:
        b = b * lr_multiplier
    if activation == 'linear' and b is not None:
        return torch.addmm(b.unsqueeze(0), x, w.t())
    return ops_ref.bias_act(x.matmul(w.t()), b, act=activation)


def mapping(sd, pre, z, num_ws, num_layers=8):
    x = z
    for i in range(num_layers):
        x = fully_connected(sd, pre + f'fc{i}.', x, activation='lrelu', lr_multiplier=0.01)
    return x.unsqueeze(1).repeat(1, num_ws, 1)


def synthesis_layer(sd, pre, x, w, up, f):
    styles = fully_connected(sd, pre + 'affine.', w)
    x = ops_ref.modulated_conv2d(x, sd[pre + 'weight'], styles, up=up, padding=1, resample_filter=f, flip_weight=(up == 1))
    return ops_ref.bias_act(x, sd[pre + 'bias'], act='lrelu')


def torgb_layer(sd, pre, x, w):
    weight = sd[pre + 'weight']
    styles = fully_connected(sd, pre + 'affine.', w) * (1.0 / math.sqrt(weight.shape[1] * weight.shape[2] ** 2))
    x = ops_ref.modulated_conv2d(x, weight, styles, demodulate=False)
    return ops_ref.bias_act(x, sd[pre + 'bias'])


def block_resolutions(img_resolution):
    return [2 ** i for i in range(2, int(math.log2(img_resolution)) + 1)]


def num_ws(img_resolution):
    return 2 * len(block_resolutions(img_resolution))  # 1 + 2*(n-1) convs + last torgb


def synthesis(sd, pre, ws, img_resolution):
    f = ops_ref.setup_filter([1, 3, 3, 1])
    x = img = None
    w_idx = 0
    for res in block_resolutions(img_resolution):
        p = pre + f'b{res}.'
        nconv = 1 if res == 4 else 2
        cur = ws[:, w_idx: w_idx + nconv + 1]
        w_idx += nconv
        k = 0
        if res == 4:
            x = sd[p + 'const'].unsqueeze(0).repeat(ws.shape[0], 1, 1, 1)
        else:
            x = synthesis_layer(sd, p + 'conv0.', x, cur[:, k], 2, f); k += 1
        x = synthesis_layer(sd, p + 'conv1.', x, cur[:, k], 1, f); k += 1
        if img is not None:
            img = ops_ref.upsample2d(img, f)
        y = torgb_layer(sd, p + 'torgb.', x, cur[:, k])
        img = y if img is None else img + y
    return img


def decoder(sd, pre, z, img_resolution):
    ws = mapping(sd, pre + 'mapping.', z, num_ws(img_resolution))
    return synthesis(sd, pre + 'synthesis.', ws, img_resolution)
