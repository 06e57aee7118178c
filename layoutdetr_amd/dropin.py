"""Reference-side bindings: what a LayoutDETR maintainer adds to run the reference tree on the gfx950 hot path.

1. `bias_act_plugin` / `upfirdn2d_plugin`: objects with the EXACT pybind argument lists of the reference's CUDA plugins
   (torch_utils/ops/bias_act.cpp:33,95-98 and upfirdn2d.cpp:17,103-106), bound to the C ABI with ctypes.  The reference's own
   `torch_utils/ops/bias_act.py` / `upfirdn2d.py` keep working unchanged when `_plugin` is replaced by these
   (`install_plugins()` does it: it pre-seeds `torch_utils.custom_ops._cached_plugins`, so `custom_ops.get_plugin()` —
   custom_ops.py:62,110-112 — returns them instead of invoking nvcc).
2. `install()`: registers this package's modules under the reference's module names (`training.networks_detr`,
   `training.loss`, `training.training_loop`, `torch_utils.ops.{bias_act,upfirdn2d,conv2d_resample,conv2d_gradfix}`), so that
   `train.py`'s class paths (`train.py:202-203,263`) and `from training import training_loop` resolve here — train.py unchanged.
3. `python -m layoutdetr_amd.dropin train.py --outdir=... --gpus=8 ...`: install(), then run the script as __main__.
"""
import ctypes
import importlib
import runpy
import sys

import torch

from . import _lib


def _p(t):
    """device pointer of a tensor; an empty tensor means "absent" (bias_act.cpp:36-52)."""
    return ctypes.c_void_p(t.data_ptr()) if (t is not None and t.numel()) else None


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _same_layout(a, b):
    return a.shape == b.shape and a.stride() == b.stride()


def _dense(t):
    """Tensor::is_non_overlapping_and_dense(): some permutation of the dimensions is contiguous."""
    dims = sorted((st, sz) for st, sz in zip(t.stride(), t.shape) if sz != 1)
    run = 1
    for st, sz in dims:
        if st != run:
            return False
        run *= sz
    return True


class bias_act_plugin(object):
    """`bias_act_plugin.bias_act(x, b, xref, yref, dy, grad, dim, act, alpha, gain, clamp) -> Tensor`"""

    @staticmethod
    def bias_act(x, b, xref, yref, dy, grad, dim, act, alpha, gain, clamp):
        def chk(cond, msg):          # TORCH_CHECK -> RuntimeError
            if not cond:
                raise RuntimeError(msg)
        chk(x.is_cuda, 'x must reside on CUDA device')
        chk(x.dtype == torch.float32, 'only float32 is implemented on the gfx950 path')
        chk(b.numel() == 0 or (b.dtype == x.dtype and b.device == x.device), 'b must have the same dtype and device as x')
        for nm, t in (('xref', xref), ('yref', yref), ('dy', dy)):
            chk(t.numel() == 0 or (t.shape == x.shape and t.dtype == x.dtype and t.device == x.device), f'{nm} must have the same shape, dtype, and device as x')
            chk(t.numel() == 0 or _same_layout(t, x), f'{nm} must have the same layout as x')
        chk(x.numel() <= 2 ** 31 - 1, 'x is too large')
        chk(b.dim() == 1, 'b must have rank 1')
        chk(b.numel() == 0 or (0 <= dim < x.dim()), 'dim is out of bounds')
        chk(b.numel() == 0 or b.numel() == x.size(dim), 'b has wrong number of elements')
        chk(grad >= 0, 'grad must be non-negative')
        chk(_dense(x), 'x must be non-overlapping and dense')
        chk(b.is_contiguous(), 'b must be contiguous')
        lib = _lib.load()
        with torch.cuda.device(x.device):
            y = torch.empty_like(x)
            rc = lib.ldetr_bias_act_f32(_p(x), _p(b), _p(xref), _p(yref), _p(dy), _p(y), x.numel(), b.numel(),
                                        x.stride(dim) if b.numel() else 1, int(grad), int(act), float(alpha), float(gain), float(clamp), _stream())
        _lib.check(rc, 'bias_act')
        return y


class upfirdn2d_plugin(object):
    """`upfirdn2d_plugin.upfirdn2d(x, f, upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip, gain) -> Tensor`"""

    @staticmethod
    def upfirdn2d(x, f, upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip, gain):
        def chk(cond, msg):
            if not cond:
                raise RuntimeError(msg)
        chk(x.is_cuda, 'x must reside on CUDA device')
        chk(f.device == x.device, 'f must reside on the same device as x')
        chk(f.dtype == torch.float32, 'f must be float32')
        chk(x.dtype == torch.float32, 'only float32 is implemented on the gfx950 path')
        chk(x.numel() <= 2 ** 31 - 1 and f.numel() <= 2 ** 31 - 1, 'x / f is too large')
        chk(x.numel() > 0, 'x has zero size')
        chk(f.numel() > 0, 'f has zero size')
        chk(x.dim() == 4, 'x must be rank 4')
        chk(f.dim() == 2, 'f must be rank 2')
        chk(f.size(0) >= 1 and f.size(1) >= 1, 'f must be at least 1x1')
        chk(upx >= 1 and upy >= 1, 'upsampling factor must be at least 1')
        chk(downx >= 1 and downy >= 1, 'downsampling factor must be at least 1')
        N, C, H, W = x.shape
        outW = (W * upx + padx0 + padx1 - f.size(1) + downx) // downx
        outH = (H * upy + pady0 + pady1 - f.size(0) + downy) // downy
        chk(outW >= 1 and outH >= 1, 'output must be at least 1x1')
        cl = x.stride(1) == 1 and C > 1                      # x.suggest_memory_format() (upfirdn2d.cpp:38)
        lib = _lib.load()
        with torch.cuda.device(x.device):
            y = torch.empty((N, C, outH, outW), device=x.device, dtype=x.dtype, memory_format=torch.channels_last if cl else torch.contiguous_format)
            xs = (ctypes.c_int64 * 4)(*x.stride()); ys = (ctypes.c_int64 * 4)(*y.stride())
            rc = lib.ldetr_upfirdn2d_f32(_p(x), _p(f), _p(y), N, C, H, W, xs, f.size(0), f.size(1), f.stride(0), f.stride(1), int(upx), int(upy),
                                         int(downx), int(downy), int(padx0), int(padx1), int(pady0), int(pady1), 1 if flip else 0, float(gain),
                                         outH, outW, ys, None, 0, 0.0, 1.0, _stream())
        _lib.check(rc, 'upfirdn2d')
        return y


_ALIASES = {
    'training.networks_detr': 'layoutdetr_amd.training.networks_detr',
    'training.loss': 'layoutdetr_amd.training.loss',
    'training.training_loop': 'layoutdetr_amd.training.training_loop',
    'training.detr_transformer': 'layoutdetr_amd.training.detr_transformer',
    'training.detr_backbone': 'layoutdetr_amd.training.detr_backbone',
    'training.networks_stylegan2': 'layoutdetr_amd.training.networks_stylegan2',
    'training.dataset_layoutganpp': 'layoutdetr_amd.training.dataset_layoutganpp',     # train.py:107 names the dataset class through this module
    'torch_utils.ops.bias_act': 'layoutdetr_amd.torch_utils.ops.bias_act',
    'torch_utils.ops.upfirdn2d': 'layoutdetr_amd.torch_utils.ops.upfirdn2d',
    'torch_utils.ops.conv2d_resample': 'layoutdetr_amd.torch_utils.ops.conv2d_resample',
    'torch_utils.ops.conv2d_gradfix': 'layoutdetr_amd.torch_utils.ops.conv2d_gradfix',
}


def install(modules=None):
    """Make `import training.networks_detr` (etc.) resolve to this package.  Call before the reference imports them."""
    out = {}
    for ref_name, here in _ALIASES.items():
        if modules is not None and ref_name not in modules:
            continue
        mod = importlib.import_module(here)
        sys.modules[ref_name] = mod
        parent, _, leaf = ref_name.rpartition('.')
        if parent in sys.modules:                  # `from training import training_loop` looks the attribute up on the package
            setattr(sys.modules[parent], leaf, mod)
        out[ref_name] = mod
    if 'training.networks_detr' in out:
        # under the reference's driver an unspecified text_mode means what the reference always builds: tokenizer + text encoder + LM decoder
        out['training.networks_detr'].REFERENCE_DEFAULTS = True
    return out


def uninstall():
    """Undo install(): drop the aliases this module registered and restore the stand-alone constructor defaults (tests)."""
    for ref_name, here in _ALIASES.items():
        mod = sys.modules.get(ref_name)
        if mod is not None and mod.__name__ == here:
            del sys.modules[ref_name]
            parent, _, leaf = ref_name.rpartition('.')
            if parent in sys.modules and getattr(sys.modules[parent], leaf, None) is mod:
                delattr(sys.modules[parent], leaf)
    nd = sys.modules.get('layoutdetr_amd.training.networks_detr')
    if nd is not None:
        nd.REFERENCE_DEFAULTS = False


def install_plugins():
    """Keep the reference's own torch_utils/ops/{bias_act,upfirdn2d}.py and swap only the native plugins under them."""
    custom_ops = importlib.import_module('torch_utils.custom_ops')
    custom_ops._cached_plugins['bias_act_plugin'] = bias_act_plugin
    custom_ops._cached_plugins['upfirdn2d_plugin'] = upfirdn2d_plugin
    return custom_ops._cached_plugins


def main(argv):
    if not argv:
        raise SystemExit('usage: python -m layoutdetr_amd.dropin <script.py> [script args...]')
    import os
    sys.path.insert(0, os.path.dirname(os.path.abspath(argv[0])))
    for pkg in ('training', 'torch_utils', 'torch_utils.ops'):     # import the reference's packages first so aliases attach to them
        try:
            importlib.import_module(pkg)
        except ImportError:
            pass
    install()
    sys.argv = list(argv)
    runpy.run_path(argv[0], run_name='__main__')


if __name__ == '__main__':
    main(sys.argv[1:])
