"""The reference's dataset (training/dataset_layoutganpp.py) for the gfx950 hot path: zip + `non_image.json` reader.

On-disk format (written by the reference's dataset_tool.py:295-366, read by training/dataset_layoutganpp.py:214-342): one zip
holding `non_image.json` = {"samples": [[base_fname, {bboxes [n][4] (xc, yc, w, h in page units), labels [n], texts [n],
page_label, attr {name, width, height, num_bbox_labels}}], ...]} and per sample `<base>_background_orig.png`,
`<base>_<i>_patch.png`, `<base>_<i>_patch_orig.png`, `<base>_<i>_patch_mask.png` (i < n <= 9).  Every field is padded to
nine slots with a validity mask (`to_dense_batch`, :29-41).

`LayoutDataset(path, background_size=..., mode=...)`, same constructor arguments / properties / item keys as the reference:

* mode='device' (default, the training path): `__getitem__` inflates ONLY the page background PNG (byte-serial entropy
  decoding stays on the host) and ships it as it is, uint8 [H, W, 3] (3 bytes per pixel instead of 12); resize to
  `background_size` + ImageNet normalisation + HWC -> CHW run on the GPU for the whole batch (`background_to_tensor`, two
  launches, bit-identical to Pillow's Lanczos and the reference's fp32 lines :330-338).  The 9 x 3 patch PNGs per sample
  (:283-327) are NEVER opened: `networks_detr` reads `bbox_patch` for its shape only (networks_detr.py:133-187), so 'patches' is
  a 0-stride zero view of the right shape; 'patches_orig' / 'patch_masks' / 'background_orig' (snapshot image grids only) are absent.
  `LayoutDataset.collate` keeps the pages uint8 and the patch placeholder 0-stride (torch's default collate would materialise 7 MB
  of zeros per sample); `training_loop()` passes it to the DataLoader and calls `batch_backgrounds_to_device`.
* mode='reference': every key of the reference's item, computed on the host exactly as the reference does (PIL decode + Lanczos
  resize, float normalise, transpose) — what `setup_snapshot` (training_loop.py:36-59) needs, and the side that
  tests/golden/dataset.npz (the reference's own `__getitem__` on tests/golden/dataset_tiny.zip) pins.

Device side (`background_to_tensor`, `csrc/resample.hip`): there is no CPU fallback — without the HIP library or a GPU tensor it raises.
"""
import ctypes
import json
import os
import zipfile

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


# ---------------------------------------------------------------------------------------------------------------------------------
# host side: the zip + non_image.json reader


def to_dense_batch(data, is_str=False):
    """dataset_layoutganpp.py:29-41: pad the leading (element) axis to nine slots; mask = which slots hold an element."""
    n = len(data)
    if n > 9:
        raise ValueError(f'to_dense_batch: {n} elements, the format holds at most 9 per layout')
    mask = np.arange(9) < n
    if is_str:
        return list(data) + [''] * (9 - n), mask
    data = np.asarray(data)
    out = np.zeros((9,) + data.shape[1:], dtype=data.dtype)
    out[:n] = data
    return out, mask


def _patch_canvas_size(width, height):
    """(:287-292) the longer side becomes 256, the other keeps the aspect ratio, rounded down to an even number."""
    if width > height:
        return 256, int(float(height) / float(width) * 256.0) // 2 * 2
    return int(float(width) / float(height) * 256.0) // 2 * 2, 256


class Dataset(torch.utils.data.Dataset):
    """Base class with the reference's constructor arguments and properties (dataset_layoutganpp.py:45-210)."""

    def __init__(self, name, raw_shape, num_bbox_labels, max_size=None, use_labels=False, background_size=1024, random_seed=0):
        self._name = name
        self._raw_shape = list(raw_shape)
        self._num_bbox_labels = num_bbox_labels
        self._colors = None
        self._use_labels = use_labels
        self.background_size = background_size
        self._raw_labels = None
        self._label_shape = None
        self._raw_idx = np.arange(self._raw_shape[0], dtype=np.int64)
        if (max_size is not None) and (self._raw_idx.size > max_size):       # :66-69: a seeded subset, kept in file order
            np.random.RandomState(random_seed).shuffle(self._raw_idx)
            self._raw_idx = np.sort(self._raw_idx[:max_size])

    def _get_raw_labels(self):
        if self._raw_labels is None:
            self._raw_labels = self._load_raw_labels() if self._use_labels else None
            if self._raw_labels is None:
                self._raw_labels = np.zeros([self._raw_shape[0], 0], dtype=np.float32)
            assert isinstance(self._raw_labels, np.ndarray) and self._raw_labels.shape[0] == self._raw_shape[0]
            assert self._raw_labels.dtype in [np.float32, np.int64]
            if self._raw_labels.dtype == np.int64:
                assert self._raw_labels.ndim == 1 and np.all(self._raw_labels >= 0)
        return self._raw_labels

    def close(self):
        pass

    def _load_raw_data(self, raw_idx):
        raise NotImplementedError

    def _load_raw_labels(self):
        raise NotImplementedError

    def __getstate__(self):
        return dict(self.__dict__, _raw_labels=None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return self._raw_idx.size

    def __getitem__(self, idx):
        return self._load_raw_data(self._raw_idx[idx]), self.get_label(idx)

    def get_label(self, idx):
        label = self._get_raw_labels()[self._raw_idx[idx]]
        if label.dtype == np.int64:
            onehot = np.zeros(self.label_shape, dtype=np.float32)
            onehot[label] = 1
            label = onehot
        return label.copy()

    def get_details(self, idx):
        raw_idx = int(self._raw_idx[idx])
        return dict(raw_idx=raw_idx, raw_label=self._get_raw_labels()[raw_idx].copy())

    name = property(lambda self: self._name)
    patch_shape = property(lambda self: list(self._raw_shape[1:]))
    num_assets = property(lambda self: self.patch_shape[0])
    num_channels = property(lambda self: self.patch_shape[1])
    height = property(lambda self: self.patch_shape[2])
    width = property(lambda self: self.patch_shape[3])
    background_size_for_training = property(lambda self: self.background_size)
    num_bbox_labels = property(lambda self: self._num_bbox_labels)

    @property
    def colors(self):
        """One RGB triple per bbox label for the snapshot grids (:183-188 takes them from seaborn's 'husl' palette; seaborn is not a
        dependency of the hot path: evenly spaced hues at fixed lightness / saturation instead — drawing only, never a network input)."""
        if self._colors is None:
            import colorsys
            n = self._num_bbox_labels
            self._colors = [tuple(int(v * 255) for v in colorsys.hls_to_rgb(i / max(n, 1), 0.65, 0.9)) for i in range(n)]
        return self._colors

    @property
    def label_shape(self):
        if self._label_shape is None:
            raw = self._get_raw_labels()
            self._label_shape = [int(np.max(raw)) + 1] if raw.dtype == np.int64 else raw.shape[1:]
        return list(self._label_shape)

    @property
    def label_dim(self):
        assert len(self.label_shape) == 1
        return self.label_shape[0]

    has_labels = property(lambda self: any(x != 0 for x in self.label_shape))
    has_onehot_labels = property(lambda self: self._get_raw_labels().dtype == np.int64)


class LayoutDataset(Dataset):
    """dataset_layoutganpp.py:214-352.  See the module docstring for `mode`."""

    def __init__(self, path, xflip=False, background_size=1024, mode='device', **super_kwargs):
        if mode not in ('device', 'reference'):
            raise ValueError("LayoutDataset: mode must be 'device' or 'reference'")
        self._path = path
        self.background_size = background_size
        self.mode = mode
        self._zipfile = None
        if os.path.splitext(self._path)[1].lower() != '.zip':
            raise IOError('Path must point to a zip')                        # :229
        self._type = 'zip'
        self._all_fnames = set(self._get_zipfile().namelist())
        if 'non_image.json' not in self._all_fnames:
            raise IOError(f'{path}: no non_image.json in the archive')
        with self._open_file('non_image.json') as f:
            self._samples = json.load(f)['samples']
        if not self._samples:
            raise IOError(f'{path}: non_image.json lists no samples')
        parts = self._path.split('/')
        name = parts[-3] if len(parts) >= 3 else os.path.splitext(os.path.basename(self._path))[0]      # :237
        raw_shape = [len(self._samples)] + self._patch_orig_shape(0)
        num_bbox_labels = self._samples[0][1]['attr']['num_bbox_labels']
        super().__init__(name=name, raw_shape=raw_shape, num_bbox_labels=num_bbox_labels, background_size=background_size, **super_kwargs)

    def _get_zipfile(self):
        if self._zipfile is None:
            self._zipfile = zipfile.ZipFile(self._path)
        return self._zipfile

    def _open_file(self, fname):
        return self._get_zipfile().open(fname, 'r')

    def close(self):
        try:
            if self._zipfile is not None:
                self._zipfile.close()
        finally:
            self._zipfile = None

    def __getstate__(self):                      # DataLoader workers re-open the archive themselves
        return dict(super().__getstate__(), _zipfile=None)

    def _patch_orig_shape(self, raw_idx):
        """[9, 3, H, W] of sample `raw_idx`'s '<base>_0_patch_orig.png' — the reference decodes the whole sample to learn this (:238);
        the PNG header is enough."""
        import PIL.Image
        base = self._samples[raw_idx][0]
        with self._open_file(base + '_0_patch_orig.png') as f:
            im = PIL.Image.open(f)
            w, h = im.size
        return [9, 3, h, w]

    def _decode(self, fname):
        import PIL.Image
        with self._open_file(fname) as f:
            im = PIL.Image.open(f)
            im.load()
        return im

    def _fields(self, raw_idx):
        """The non-image fields of one sample (:270-281), padded to nine slots."""
        meta = self._samples[raw_idx][1]
        bboxes = np.array(meta['bboxes'])
        bboxes_batch, mask = to_dense_batch(bboxes.reshape(-1, 4) if bboxes.size else np.zeros((0, 4)))
        labels_batch, _ = to_dense_batch(np.array(meta['labels']))
        texts_batch, _ = to_dense_batch(meta['texts'], is_str=True)
        return dict(name=meta['attr']['name'], W_page=meta['attr']['width'], H_page=meta['attr']['height'],
                    bboxes=bboxes_batch.astype(np.float32), labels=labels_batch.astype(np.int64), texts=texts_batch, mask=mask), int(mask.sum())

    def _load_raw_data(self, raw_idx):
        raw_idx = int(raw_idx)
        out, n = self._fields(raw_idx)
        base = self._samples[raw_idx][0]
        if self.mode == 'device':
            page = np.array(self._decode(base + '_background_orig.png'))
            if page.ndim != 3 or page.shape[2] != 3:
                raise ValueError(f'{base}_background_orig.png: expected an RGB page (:334)')
            out['background'] = page                                                          # uint8 [H, W, 3]; resized + normalised on the GPU
            out['patches'] = np.broadcast_to(np.zeros((), np.float32), (9, 3, 256, 256))      # shape only, never read
            return out
        import PIL.Image
        lanczos = PIL.Image.LANCZOS                                                           # = the reference's PIL.Image.ANTIALIAS (alias removed in Pillow 10)
        mean = np.reshape(np.array(RGB_MEAN).astype(np.float32), (1, 1, 3))
        std = np.reshape(np.array(RGB_STD).astype(np.float32), (1, 1, 3))

        def norm_chw(a):
            return ((a.astype(np.float32) / 255.0 - mean) / std).transpose(2, 0, 1)

        patches, patches_orig, patch_masks = [], [], []
        for i in range(n):
            im = self._decode(base + '_%d_patch.png' % i)                                    # :283-303
            wn, hn = _patch_canvas_size(im.width, im.height)
            small = np.array(im.resize((wn, hn), lanczos))
            assert small.ndim == 3 and small.shape[2] == 3
            canvas = np.zeros((256, 256, 3), np.float32)
            canvas[128 - hn // 2:128 + hn // 2, 128 - wn // 2:128 + wn // 2] = (small.astype(np.float32) / 255.0 - mean) / std
            patches.append(canvas.transpose(2, 0, 1))
            po = np.array(self._decode(base + '_%d_patch_orig.png' % i))                     # :305-315
            assert po.ndim == 3 and po.shape[2] == 3
            patches_orig.append(norm_chw(po))
            pm = np.array(self._decode(base + '_%d_patch_mask.png' % i))[:, :, np.newaxis]   # :317-327
            patch_masks.append((pm.astype(np.float32) / 255.0).transpose(2, 0, 1))
        out['patches'] = to_dense_batch(np.stack(patches, axis=0))[0]
        out['patches_orig'] = to_dense_batch(np.stack(patches_orig, axis=0))[0]
        out['patch_masks'] = to_dense_batch(np.stack(patch_masks, axis=0))[0]
        page = self._decode(base + '_background_orig.png')                                   # :329-340
        small = np.array(page.resize((self.background_size, self.background_size), lanczos))
        assert small.ndim == 3 and small.shape[2] == 3
        out['background'] = norm_chw(small)
        out['background_orig'] = norm_chw(np.array(page))
        return out

    def _load_raw_labels(self):
        """:344-352: the reference collects `page_label` and then returns None on every path (the conversion lines are commented out),
        so `use_labels=True` still yields a zero-width label: c_dim = 0 in every run `train.py` configures."""
        return None

    @staticmethod
    def collate(items):
        """DataLoader collate for mode='device' items: numeric fields stacked as torch's default collate does, 'texts' transposed to
        nine lists of B strings (what the default collate makes of a list of strings per item; training_loop.py:259 transposes it back),
        backgrounds kept uint8 — one [B, H, W, 3] tensor when the pages share a size, else a list — and 'patches' a 0-stride view."""
        samples, labels = zip(*items)
        b = len(samples)
        out = {}
        for k in ('bboxes', 'labels', 'mask'):
            out[k] = torch.from_numpy(np.stack([s[k] for s in samples], axis=0))
        for k in ('W_page', 'H_page'):
            out[k] = torch.tensor([s[k] for s in samples])
        out['name'] = [s['name'] for s in samples]
        out['texts'] = [tuple(s['texts'][j] for s in samples) for j in range(9)]
        pages = [s['background'] for s in samples]
        if pages[0].dtype == np.uint8:
            same = all(p.shape == pages[0].shape for p in pages)
            out['background'] = torch.from_numpy(np.stack(pages, 0)) if same else [torch.from_numpy(np.ascontiguousarray(p)) for p in pages]
        else:
            out['background'] = torch.from_numpy(np.stack(pages, 0))
        p0 = samples[0]['patches']
        if 0 in p0.strides:
            out['patches'] = torch.zeros((1,) * (p0.ndim + 1)).expand(b, *p0.shape)
        else:
            out['patches'] = torch.from_numpy(np.stack([s['patches'] for s in samples], 0))
        for k in ('patches_orig', 'patch_masks', 'background_orig'):
            if k in samples[0]:
                out[k] = torch.from_numpy(np.stack([s[k] for s in samples], 0))
        return out, torch.from_numpy(np.stack(labels, 0))


def batch_backgrounds_to_device(background, background_size, device):
    """'background' of a collated batch -> float32 [B, 3, S, S] on `device`.  uint8 pages (mode='device': one [B, H, W, 3] tensor, or a
    list when page sizes differ) are uploaded as bytes and resized + normalised by `background_to_tensor`, one call per distinct page
    size; float input (mode='reference' / other datasets) is what the reference already hands over (training_loop.py:264)."""
    if torch.is_tensor(background):
        if background.dtype != torch.uint8:
            return background.to(device).float()
        return background_to_tensor(background.to(device, non_blocking=True), background_size)
    groups = {}
    for i, p in enumerate(background):
        groups.setdefault(tuple(p.shape), []).append(i)
    out = torch.empty((len(background), 3, background_size, background_size), dtype=torch.float32, device=device)
    for idx in groups.values():
        out[torch.tensor(idx, device=device)] = background_to_tensor(torch.stack([background[i] for i in idx]).to(device, non_blocking=True), background_size)
    return out


def patch_placeholder_to_device(patches, device):
    """'patches' of a collated batch on `device` without materialising a 0-stride placeholder (`.to()` would: 7 MB per sample)."""
    if patches.stride(0) == 0:
        return torch.zeros((1,) * patches.ndim, dtype=patches.dtype, device=device).expand(patches.shape)
    return patches.to(device)
