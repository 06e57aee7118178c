"""Device side of the reference's dataset item loader, background branch.

Reference (training/dataset_layoutganpp.py:330-338): every `__getitem__` decodes `<name>_background_orig.png` with PIL,
resizes the (typically 1024 x 1024) page to `background_size` with `PIL.Image.ANTIALIAS`, converts to float, normalises with the
ImageNet mean/std and transposes to CHW — on the DataLoader workers' CPU cores, per sample.  (It also decodes and resizes up to
9 x 3 patch PNGs per sample, `:283-327`, which `networks_detr` never consumes: `bbox_patch` is used for its shape only,
networks_detr.py:133-187.)

Here the PNG inflate stays on the host (byte-serial entropy decoding), the decoded uint8 pages go to HBM as they are (3 bytes
per pixel instead of 12), and resize + normalise + transpose run as two HIP kernels over the whole batch
(`csrc/resample.hip`): bit-identical to Pillow's Lanczos resize and to the reference's fp32 normalisation
(tests/test_kernels_gpu.py::test_background_resize_normalize_*; oracle/resample_ref.py pinned to Pillow by tests/golden/resample.npz).
There is no CPU fallback: without the HIP library or a GPU tensor this raises.
"""
import ctypes

import numpy as np
import torch

from ..hip import core

RGB_MEAN = (0.485, 0.456, 0.406)   # dataset_layoutganpp.py:268
RGB_STD = (0.229, 0.224, 0.225)    # dataset_layoutganpp.py:269

_coeff_cache = {}


def resample_coeffs(in_size, out_size, device):
    """(bounds [out, 2] int32, weights [ksize, out] int32, ksize) of Pillow's Lanczos window on `device`, computed once per
    (in, out) on the host by the C-ABI library and cached."""
    key = (int(in_size), int(out_size), str(device))
    hit = _coeff_cache.get(key)
    if hit is not None:
        return hit
    lib = core.lib()
    ks = ctypes.c_int(0)
    core.check(lib.ldetr_resample_coeffs(in_size, out_size, None, None, 0, ctypes.byref(ks)), 'resample_coeffs')
    bounds = np.zeros((out_size, 2), np.int32)
    weights = np.zeros((ks.value, out_size), np.int32)
    core.check(lib.ldetr_resample_coeffs(in_size, out_size, bounds.ctypes.data_as(ctypes.c_void_p), weights.ctypes.data_as(ctypes.c_void_p),
                                         weights.size, ctypes.byref(ks)), 'resample_coeffs')
    out = (torch.from_numpy(bounds).to(device), torch.from_numpy(weights).to(device), ks.value)
    _coeff_cache[key] = out
    return out


def background_to_tensor(pages_u8, background_size, return_u8=False, mean=RGB_MEAN, std=RGB_STD):
    """Decoded pages, uint8 [n, H, W, 3] (or [H, W, 3]) in GPU memory -> float32 [n, 3, S, S]: what `_load_raw_data` returns under
    'background', for the whole batch in two launches.  return_u8 additionally returns the resized uint8 pages [n, S, S, 3]."""
    core.require_gpu(pages_u8)
    single = pages_u8.ndim == 3
    if single:
        pages_u8 = pages_u8[None]
    if pages_u8.dtype != torch.uint8 or pages_u8.ndim != 4 or pages_u8.shape[-1] != 3:
        raise ValueError('background_to_tensor: expected uint8 [n, H, W, 3] (the reference asserts 3 channels, dataset_layoutganpp.py:334)')
    pages_u8 = pages_u8.contiguous()
    n, H, W, _ = pages_u8.shape
    S = int(background_size)
    dev = pages_u8.device
    hb, hk, hks = resample_coeffs(W, S, dev)
    vb, vk, vks = resample_coeffs(H, S, dev)
    tmp = torch.empty((n, H, S, 3), dtype=torch.uint8, device=dev)
    out = torch.empty((n, 3, S, S), dtype=torch.float32, device=dev)
    u8 = torch.empty((n, S, S, 3), dtype=torch.uint8, device=dev) if return_u8 else None
    m = [float(np.float32(v)) for v in mean]
    s = [float(np.float32(v)) for v in std]
    core.check(core.lib().ldetr_resize_normalize_u8(core.ptr(pages_u8), n, H, W, S, S, core.ptr(hb), core.ptr(hk), hks, core.ptr(vb), core.ptr(vk), vks,
                                                    core.ptr(tmp), core.ptr(u8) if u8 is not None else None, core.ptr(out),
                                                    m[0], m[1], m[2], s[0], s[1], s[2], core.stream()), 'resize_normalize_u8')
    if single:
        out = out[0]
        u8 = u8[0] if u8 is not None else None
    return (out, u8) if return_u8 else out
