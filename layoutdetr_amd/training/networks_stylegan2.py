"""StyleGAN2 `Decoder` chain (D's background-reconstruction head) on the fused gfx950 layers.

Interface of the reference `training/networks_stylegan2.py` for the classes the hot path constructs
(networks_detr.py:261): FullyConnectedLayer :92-126, SynthesisLayer :272-331, ToRGBLayer :336-356,
SynthesisBlock :361-460, SynthesisNetwork :465-520, DecoderMappingNetwork :903-967, Decoder :972-994 —
same constructor arguments, parameter names and shapes (state_dict-compatible), same outputs.
Only the configuration LayoutDETR uses is implemented (architecture='skip', use_noise=False,
num_fp16_res=0, conv_clamp=None, non-fused modconv); anything else raises NotImplementedError.
The off-path classes of the reference file (MappingNetwork, Generator, Discriminator*, Encoder*) are
out of scope.
"""
import math

import numpy as np
import torch

from ..hip import core, modconv
from ..hip.linear import linear
from ..torch_utils.ops import bias_act, upfirdn2d


def normalize_2nd_moment(x, dim=1, eps=1e-8):
    return x * (x.square().mean(dim=dim, keepdim=True) + eps).rsqrt()


class FullyConnectedLayer(torch.nn.Module):
    def __init__(self, in_features, out_features, bias=True, activation='linear', lr_multiplier=1, bias_init=0):
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.activation = activation
        self.weight = torch.nn.Parameter(torch.randn([out_features, in_features]) / lr_multiplier)
        self.bias = torch.nn.Parameter(torch.full([out_features], np.float32(bias_init))) if bias else None
        self.weight_gain = lr_multiplier / np.sqrt(in_features)
        self.bias_gain = lr_multiplier

    def forward(self, x):
        b = self.bias
        spec = bias_act.activation_funcs[self.activation]
        if self.activation == 'lrelu' and b is not None and self.bias_gain != 1 and self.bias_gain > 0:
            # act(x (w g_w)^T + b g_b) gain = lrelu(g_b (x (w g_w / g_b)^T + b)) gain = (g_b gain) lrelu(x (w g_w / g_b)^T + b)  (lrelu is positively
            # homogeneous): the bias gain of the mapping network's layers (lr_multiplier 0.01, networks_stylegan2.py:108-111) moves into the GEMM's scalars
            # instead of a `b * bias_gain` launch forward and its gradient launch backward per layer
            return linear(x, self.weight, b, act=core.ACT_LRELU, act_alpha=float(spec.def_alpha), act_gain=float(spec.def_gain * self.bias_gain),
                          wscale=float(self.weight_gain / self.bias_gain))
        if b is not None and self.bias_gain != 1:
            b = b * self.bias_gain
        if self.activation == 'linear':
            return linear(x, self.weight, b, wscale=float(self.weight_gain))
        if self.activation in ('relu', 'lrelu'):
            act = core.ACT_RELU if self.activation == 'relu' else core.ACT_LRELU
            return linear(x, self.weight, b, act=act, act_alpha=float(spec.def_alpha), act_gain=float(spec.def_gain),
                          wscale=float(self.weight_gain))
        y = linear(x, self.weight, None, wscale=float(self.weight_gain))
        return bias_act.bias_act(y, b, act=self.activation)



class SynthesisLayer(torch.nn.Module):
    def __init__(self, in_channels, out_channels, w_dim, resolution, kernel_size=3, up=1, use_noise=True, activation='lrelu',
                 resample_filter=[1, 3, 3, 1], conv_clamp=None, channels_last=False):
        super().__init__()
        if use_noise or conv_clamp is not None or activation != 'lrelu' or kernel_size != 3 or up not in (1, 2):
            raise NotImplementedError('SynthesisLayer: only use_noise=False, conv_clamp=None, lrelu, 3x3, up in {1,2}')
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.w_dim = w_dim
        self.resolution = resolution
        self.up = up
        self.use_noise = use_noise
        self.activation = activation
        self.conv_clamp = conv_clamp
        self.register_buffer('resample_filter', upfirdn2d.setup_filter(resample_filter))
        self.padding = kernel_size // 2
        self.act_gain = bias_act.activation_funcs[activation].def_gain
        self.affine = FullyConnectedLayer(w_dim, in_channels, bias_init=1)
        self.weight = torch.nn.Parameter(torch.randn([out_channels, in_channels, kernel_size, kernel_size]).to(memory_format=torch.channels_last))
        self.bias = torch.nn.Parameter(torch.zeros([out_channels]))

    def forward(self, x, w, noise_mode='none', fused_modconv=False, gain=1):
        """x: NHWC [B, r, r, Cin] -> NHWC [B, r*up, r*up, Cout]."""
        if fused_modconv:
            raise NotImplementedError('fused_modconv=True is not used by LayoutDETR (networks_detr.py:261)')
        styles = self.affine(w)
        g = float(self.act_gain * gain)
        if self.up == 1:
            return modconv.modconv3x3(x, self.weight, styles, self.bias, 0.2, g)
        return modconv.modconv3x3_up2(x, self.weight, styles, self.bias, self.resample_filter, 0.2, g)

    def extra_repr(self):
        return (f'in_channels={self.in_channels:d}, out_channels={self.out_channels:d}, w_dim={self.w_dim:d}, '
                f'resolution={self.resolution:d}, up={self.up}, activation={self.activation:s}')


class ToRGBLayer(torch.nn.Module):
    def __init__(self, in_channels, out_channels, w_dim, kernel_size=1, conv_clamp=None, channels_last=False):
        super().__init__()
        if kernel_size != 1 or conv_clamp is not None:
            raise NotImplementedError('ToRGBLayer: only 1x1 without clamping')
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.w_dim = w_dim
        self.conv_clamp = conv_clamp
        self.affine = FullyConnectedLayer(w_dim, in_channels, bias_init=1)
        self.weight = torch.nn.Parameter(torch.randn([out_channels, in_channels, kernel_size, kernel_size]))
        self.bias = torch.nn.Parameter(torch.zeros([out_channels]))
        self.weight_gain = 1 / np.sqrt(in_channels * (kernel_size ** 2))

    def forward(self, x, w, fused_modconv=False):
        styles = self.affine(w) * float(self.weight_gain)
        return modconv.torgb(x, self.weight, styles, self.bias)



class SynthesisBlock(torch.nn.Module):
    def __init__(self, in_channels, out_channels, w_dim, resolution, img_channels, is_last, architecture='skip',
                 resample_filter=[1, 3, 3, 1], conv_clamp=256, use_fp16=False, fp16_channels_last=False,
                 fused_modconv_default=True, **layer_kwargs):
        super().__init__()
        if architecture != 'skip' or use_fp16:
            raise NotImplementedError("SynthesisBlock: only architecture='skip' in fp32")
        self.in_channels = in_channels
        self.w_dim = w_dim
        self.resolution = resolution
        self.img_channels = img_channels
        self.is_last = is_last
        self.architecture = architecture
        self.use_fp16 = use_fp16
        self.channels_last = False
        self.fused_modconv_default = fused_modconv_default
        self.register_buffer('resample_filter', upfirdn2d.setup_filter(resample_filter))
        self.num_conv = 0
        self.num_torgb = 0
        if in_channels == 0:
            self.const = torch.nn.Parameter(torch.randn([out_channels, resolution, resolution]))
        if in_channels != 0:
            self.conv0 = SynthesisLayer(in_channels, out_channels, w_dim=w_dim, resolution=resolution, up=2,
                                        resample_filter=resample_filter, conv_clamp=conv_clamp, **layer_kwargs)
            self.num_conv += 1
        self.conv1 = SynthesisLayer(out_channels, out_channels, w_dim=w_dim, resolution=resolution, conv_clamp=conv_clamp, **layer_kwargs)
        self.num_conv += 1
        self.torgb = ToRGBLayer(out_channels, img_channels, w_dim=w_dim, conv_clamp=conv_clamp)
        self.num_torgb += 1

    def forward(self, x, img, ws, force_fp32=False, fused_modconv=None, update_emas=False, **layer_kwargs):
        """x: NHWC features (or None for the first block); img: NHWC [B, r/2, r/2, 3] (or None)."""
        w_iter = iter(ws.unbind(dim=1))
        if fused_modconv is None:
            fused_modconv = self.fused_modconv_default
        if fused_modconv == 'inference_only':
            fused_modconv = not self.training
        if self.in_channels == 0:
            x = self.const.permute(1, 2, 0).unsqueeze(0).repeat([ws.shape[0], 1, 1, 1])
            x = self.conv1(x, next(w_iter), fused_modconv=fused_modconv, **layer_kwargs)
        else:
            x = self.conv0(x, next(w_iter), fused_modconv=fused_modconv, **layer_kwargs)
            x = self.conv1(x, next(w_iter), fused_modconv=fused_modconv, **layer_kwargs)
        if img is not None:
            img = upfirdn2d.upsample2d(img.permute(0, 3, 1, 2), self.resample_filter).permute(0, 2, 3, 1)
        y = self.torgb(x, next(w_iter), fused_modconv=fused_modconv)
        img = img + y if img is not None else y
        return x, img



class SynthesisNetwork(torch.nn.Module):
    def __init__(self, w_dim, img_resolution, img_channels, channel_base=32768, channel_max=512, num_fp16_res=4, **block_kwargs):
        assert img_resolution >= 4 and img_resolution & (img_resolution - 1) == 0
        super().__init__()
        if num_fp16_res != 0:
            raise NotImplementedError('fp16 blocks are not used by LayoutDETR (num_fp16_res=0)')
        self.w_dim = w_dim
        self.img_resolution = img_resolution
        self.img_resolution_log2 = int(np.log2(img_resolution))
        self.img_channels = img_channels
        self.num_fp16_res = num_fp16_res
        self.block_resolutions = [2 ** i for i in range(2, self.img_resolution_log2 + 1)]
        channels_dict = {res: min(channel_base // res, channel_max) for res in self.block_resolutions}
        self.num_ws = 0
        for res in self.block_resolutions:
            in_channels = channels_dict[res // 2] if res > 4 else 0
            out_channels = channels_dict[res]
            is_last = (res == self.img_resolution)
            block = SynthesisBlock(in_channels, out_channels, w_dim=w_dim, resolution=res, img_channels=img_channels,
                                   is_last=is_last, use_fp16=False, **block_kwargs)
            self.num_ws += block.num_conv
            if is_last:
                self.num_ws += block.num_torgb
            setattr(self, f'b{res}', block)

    def forward(self, ws, **block_kwargs):
        """ws: [B, num_ws, w_dim] (one latent per layer, the reference's form) or [B, w_dim] (the same latent for every layer — what
        the Decoder's mapping network produces: handed over without materialising the num_ws copies)."""
        ws = ws.to(torch.float32)
        x = img = None
        first = 0
        for res in self.block_resolutions:
            block = getattr(self, f'b{res}')
            n = block.num_conv + block.num_torgb          # the block's toRGB shares its latent with the next block's first conv
            cur = ws.unsqueeze(1).expand(-1, n, -1) if ws.ndim == 2 else ws[:, first:first + n]
            x, img = block(x, img, cur, **block_kwargs)
            first += block.num_conv
        return img.permute(0, 3, 1, 2)  # [B, 3, R, R] view (channels_last memory)


class DecoderMappingNetwork(torch.nn.Module):
    def __init__(self, z_dim, w_dim, num_ws, num_layers=8, layer_features=None, activation='lrelu', lr_multiplier=0.01,
                 w_avg_beta=0.998):
        super().__init__()
        self.z_dim = z_dim
        self.w_dim = w_dim
        self.num_ws = num_ws
        self.num_layers = num_layers
        self.w_avg_beta = w_avg_beta
        if layer_features is None:
            layer_features = w_dim
        features_list = [z_dim] + [layer_features] * (num_layers - 1) + [w_dim]
        for idx in range(num_layers):
            layer = FullyConnectedLayer(features_list[idx], features_list[idx + 1], activation=activation, lr_multiplier=lr_multiplier)
            setattr(self, f'fc{idx}', layer)
        if num_ws is not None and w_avg_beta is not None:
            self.register_buffer('w_avg', torch.zeros([w_dim]))

    def forward(self, z, truncation_psi=1, truncation_cutoff=None, update_emas=False, broadcast=True):
        """broadcast=False returns w as [B, w_dim] when every layer gets the same latent (no truncation cutoff): SynthesisNetwork takes
        that form directly.  Semantics of the reference's mapping (networks_stylegan2.py:940-964) otherwise: w_avg tracking under
        update_emas, truncation towards w_avg, optionally only for the first `truncation_cutoff` layers."""
        w = z.to(torch.float32)
        for idx in range(self.num_layers):
            w = getattr(self, f'fc{idx}')(w)
        if update_emas and self.w_avg_beta is not None:
            self.w_avg.copy_(torch.lerp(w.detach().mean(dim=0), self.w_avg, self.w_avg_beta))
        per_layer = truncation_psi != 1 and self.num_ws is not None and truncation_cutoff is not None
        if truncation_psi != 1 and not per_layer:
            assert self.w_avg_beta is not None
            w = torch.lerp(self.w_avg, w, truncation_psi)
        if self.num_ws is None or (not broadcast and not per_layer):
            return w
        ws = w.unsqueeze(1).repeat([1, self.num_ws, 1])
        if per_layer:
            assert self.w_avg_beta is not None
            head = torch.lerp(self.w_avg, ws[:, :truncation_cutoff], truncation_psi)
            ws = torch.cat([head, ws[:, truncation_cutoff:]], dim=1)
        return ws


class Decoder(torch.nn.Module):
    def __init__(self, z_dim, w_dim, img_resolution, img_channels, use_noise, mapping_kwargs={}, **synthesis_kwargs):
        super().__init__()
        self.z_dim = z_dim
        self.w_dim = w_dim
        self.img_resolution = img_resolution
        self.img_channels = img_channels
        self.synthesis = SynthesisNetwork(w_dim=w_dim, img_resolution=img_resolution, img_channels=img_channels, use_noise=use_noise,
                                          **synthesis_kwargs)
        self.num_ws = self.synthesis.num_ws
        self.mapping = DecoderMappingNetwork(z_dim=z_dim, w_dim=w_dim, num_ws=self.num_ws, **mapping_kwargs)

    def forward(self, z, truncation_psi=1, truncation_cutoff=None, update_emas=False, **synthesis_kwargs):
        ws = self.mapping(z, truncation_psi=truncation_psi, truncation_cutoff=truncation_cutoff, update_emas=update_emas, broadcast=False)
        return self.synthesis(ws, update_emas=update_emas, **synthesis_kwargs)
