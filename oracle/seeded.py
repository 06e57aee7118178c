"""ORACLE-side helper (test infrastructure only): platform-independent seeded tensors.

The composition fixtures (tests/golden/composition*.npz) pin full-size Generator / Discriminator modules
(135 M parameters) — far too large to store.  Instead every parameter is *derived* from its state-dict name
by integer arithmetic only (crc32 of the name -> splitmix64 counter stream -> top 24 bits -> uniform fp32), so
`oracle/gen_golden.py` (which drives the imported reference), the CPU oracle tests and the GPU parity tests
all rebuild bit-identical weights on any machine without relying on a library RNG.
"""
import zlib

import numpy as np
import torch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def uniform(name, shape, seed=0, lo=-1.0, hi=1.0):
    """fp32 tensor of `shape`, uniform in [lo, hi), a pure function of (name, shape, seed)."""
    n = int(np.prod(shape)) if len(shape) else 1
    base = np.uint64((zlib.crc32(name.encode()) << 32) ^ (seed * 0x9E3779B1 & 0xFFFFFFFF))
    with np.errstate(over='ignore'):
        bits = _splitmix64(np.arange(n, dtype=np.uint64) * np.uint64(0xD1342543DE82EF95) + base)
    u = (bits >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / (1 << 24))     # [0, 1), 24 bits: exact in fp32
    return torch.from_numpy((u * np.float32(hi - lo) + np.float32(lo)).reshape(shape))


def randint(name, shape, high, seed=0):
    n = int(np.prod(shape)) if len(shape) else 1
    base = np.uint64((zlib.crc32(name.encode()) << 32) ^ (seed * 0x9E3779B1 & 0xFFFFFFFF))
    with np.errstate(over='ignore'):
        bits = _splitmix64(np.arange(n, dtype=np.uint64) * np.uint64(0xD1342543DE82EF95) + base)
    return torch.from_numpy(((bits >> np.uint64(33)) % np.uint64(high)).astype(np.int64).reshape(shape))


def seeded_state_dict(module_or_sd, seed=0, skip=()):
    """name -> tensor for every floating-point entry of a state dict, scaled like a trained network:
    matrices / conv kernels ~ U(+-sqrt(3 / fan_in)) (unit-variance preserving), LayerNorm / FrozenBN scales ~ 1 +- 0.1,
    variances in [0.5, 1.5], everything else (biases, embeddings rows, tokens) ~ U(+-0.1) except embedding tables (+-0.5).
    Integer / bool buffers and names containing an entry of `skip` keep their current values."""
    sd = module_or_sd.state_dict() if hasattr(module_or_sd, 'state_dict') else module_or_sd
    out = {}
    for name in sorted(sd):
        v = sd[name]
        if not v.dtype.is_floating_point or any(s in name for s in skip):
            out[name] = v.clone()
            continue
        shape = tuple(v.shape)
        leaf = name.rsplit('.', 1)[-1]
        is_emb = any(t in name for t in ('emb_label', 'enc_text_len', 'word_embeddings', 'position_embeddings'))
        if 'resample_filter' in name or 'w_avg' in name:
            out[name] = v.clone()          # FIR taps / EMA of w: constructor values
        elif 'running_var' in name:
            out[name] = uniform(name, shape, seed, 0.5, 1.5)
        elif ('norm' in name.lower() or '.bn' in name or 'downsample.1' in name) and leaf == 'weight' and len(shape) == 1:
            out[name] = uniform(name, shape, seed, 0.9, 1.1)
        elif len(shape) >= 2 and leaf in ('weight', 'in_proj_weight') and not is_emb:
            fan_in = int(np.prod(shape[1:]))
            a = float(np.sqrt(3.0 / fan_in))
            if 'synthesis' in name or 'mapping' in name:
                a = float(np.sqrt(3.0))    # StyleGAN2 layers carry unit-variance weights and scale by 1/sqrt(fan_in) at run time
            out[name] = uniform(name, shape, seed, -a, a)
        elif is_emb and len(shape) == 2:
            out[name] = uniform(name, shape, seed, -0.5, 0.5)
        else:
            out[name] = uniform(name, shape, seed, -0.1, 0.1)
        if 'affine.bias' in name:          # StyleGAN2 style affines are initialised with bias 1 (styles around 1)
            out[name] = out[name] + 1.0
    return out


COMP_TEXTS = ['Sale', 'Up to 50% off', 'Shop now', 'New arrivals this week', 'x', 'Free shipping on orders over $25', 'Sign up', 'Limited time only!', 'ok']


def comp_inputs(B, bg, seed):
    """Inputs of the composition fixtures, a pure function of (B, bg, seed) — rebuilt by the tests (oracle/seeded.py arithmetic)."""
    xy = uniform('in.xy', (B, 9, 2), seed, 0.2, 0.8); wh = uniform('in.wh', (B, 9, 2), seed, 0.05, 0.4)
    pm = torch.zeros(B, 9, dtype=torch.bool)
    if B > 1:
        pm[1, 5:] = True
    if B > 2:
        pm[2, 1:] = True
    texts = [[COMP_TEXTS[(i * 3 + j * 5 + seed) % len(COMP_TEXTS)] for j in range(9)] for i in range(B)]
    return dict(bbox_real=torch.cat([xy, wh], -1), bbox_class=randint('in.cls', (B, 9), 8, seed), padding_mask=pm,
                background=uniform('in.bg', (B, 3, bg, bg), seed, -2.0, 2.0), texts=texts,
                z_g=uniform('in.zg', (B, 9, 4), seed, -1.7, 1.7), z_d=uniform('in.zd', (B, 9, 4), seed, -1.7, 1.7),
                feats_g=uniform('in.fg', (B, 2048, bg // 32, bg // 32), seed, 0.0, 1.5) * (uniform('in.fgm', (B, 2048, bg // 32, bg // 32), seed, 0, 1) > 0.5),
                feats_d=uniform('in.fd', (B, 2048, bg // 32, bg // 32), seed, 0.0, 1.5) * (uniform('in.fdm', (B, 2048, bg // 32, bg // 32), seed, 0, 1) > 0.5))


def grad_digest(g, samples=512):
    """((L2 norm, sum, max |.|), strided subsample) of a gradient tensor: what the composition fixtures store per parameter."""
    f = g.detach().double().flatten()
    n = f.numel()
    stride = max(n // samples, 1)
    return np.array([f.norm().item(), f.sum().item(), f.abs().max().item()], dtype=np.float64), f[::stride][:samples].numpy().copy()
