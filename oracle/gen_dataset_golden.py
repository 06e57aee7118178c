"""Generates tests/golden/dataset_tiny.zip and tests/golden/dataset.npz.

Run in the build container only:  python oracle/gen_dataset_golden.py

1. Writes a 3-sample archive in the reference's on-disk format (dataset_tool.py:295-366: `non_image.json` + per sample
   `<base>_background_orig.png`, `<base>_<i>_patch.png`, `<base>_<i>_patch_orig.png`, `<base>_<i>_patch_mask.png`) from seeded
   synthetic pixels: 9, 4 and 1 elements, 80 x 56 pages, wide / tall / square patches (both branches of the patch canvas rule).
2. Imports the REFERENCE's `training.dataset_layoutganpp.LayoutDataset` from /root/reference, reads the archive with it and stores
   what its `__getitem__` returned.  The 9 x 3 x 256 x 256 patch canvases are stored as an every-4th-pixel sub-grid plus their float64
   sums (enough to pin placement, resize and normalisation without a 7 MB fixture).
Environment drift bridged here (none of it takes part in a captured computation): `seaborn` is absent (stubbed; only `.colors`
uses it), numpy 2 dropped the `np.bool` alias the reference spells (`bool`), Pillow 10 dropped `Image.ANTIALIAS` (= `Image.LANCZOS`).
"""
import io
import json
import os
import sys
import types
import zipfile

import numpy as np
import PIL.Image

REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
BG_SIZE = 32


def _png(a):
    buf = io.BytesIO()
    PIL.Image.fromarray(a).save(buf, format='png', compress_level=0)
    return buf.getvalue()


def _smooth(rnd, h, w, c):
    """Seeded low-frequency pattern + noise (so that the Lanczos taps matter), uint8."""
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    chans = []
    for _ in range(c):
        fx, fy, ph = rnd.uniform(0.05, 0.4), rnd.uniform(0.05, 0.4), rnd.uniform(0, 6.28)
        chans.append(127.5 + 90 * np.sin(fx * xx + fy * yy + ph) + rnd.normal(0, 25, (h, w)))
    a = np.stack(chans, -1).clip(0, 255).astype(np.uint8)
    return a if c > 1 else a[..., 0]


def make_zip(path):
    rnd = np.random.RandomState(2026)
    words = ['sale', 'shop now', 'up to 50% off', 'new arrivals', 'free shipping', 'ok', 'limited time only!', 'x', 'sign up today']
    # the reference asserts background_orig.shape == patches_orig.shape[1:] == sample 0's (:121-125): one page size, page-sized patch originals
    specs = [('ads/000/page_a', 9, (80, 56), (80, 56)), ('ads/000/page_b', 4, (80, 56), (80, 56)), ('ads/001/page_c', 1, (80, 56), (80, 56))]
    patch_sizes = [(30, 12), (10, 28), (16, 16), (25, 9), (7, 7), (40, 13), (12, 33), (20, 20), (9, 31)]     # (w, h)
    samples = []
    with zipfile.ZipFile(path, 'w', zipfile.ZIP_STORED) as z:
        for base, n, (pw, ph), (ow, oh) in specs:
            xy = rnd.uniform(0.2, 0.8, (n, 2)); wh = rnd.uniform(0.05, 0.4, (n, 2))
            meta = dict(bboxes=np.concatenate([xy, wh], 1).round(6).tolist(), labels=rnd.randint(0, 8, n).tolist(), texts=[words[(i * 2 + n) % 9] for i in range(n)],
                        page_label=None, attr=dict(name=base.split('/')[-1], width=pw * 10, height=ph * 10, num_bbox_labels=8))
            samples.append([base, meta])
            z.writestr(base + '_background_orig.png', _png(_smooth(rnd, ph, pw, 3)))
            for i in range(n):
                w, h = patch_sizes[i]
                z.writestr(base + '_%d_patch.png' % i, _png(_smooth(rnd, h, w, 3)))
                z.writestr(base + '_%d_patch_orig.png' % i, _png(_smooth(rnd, oh, ow, 3)))
                z.writestr(base + '_%d_patch_mask.png' % i, _png(_smooth(rnd, oh, ow, 1)))
        z.writestr('non_image.json', json.dumps(dict(samples=samples)))


def main():
    os.makedirs(OUT, exist_ok=True)
    zpath = os.path.join(OUT, 'dataset_tiny.zip')
    make_zip(zpath)
    print('wrote', zpath, os.path.getsize(zpath), 'bytes')
    sys.modules['seaborn'] = types.ModuleType('seaborn')
    if not hasattr(np, 'bool'):
        np.bool = bool
    if not hasattr(PIL.Image, 'ANTIALIAS'):
        PIL.Image.ANTIALIAS = PIL.Image.LANCZOS
    sys.path.insert(0, REF)
    from training.dataset_layoutganpp import LayoutDataset
    ds = LayoutDataset(path=zpath, background_size=BG_SIZE, use_labels=False, max_size=None, xflip=False)
    d = dict(background_size=BG_SIZE, n=len(ds), name=ds.name, patch_shape=np.array(ds.patch_shape), num_bbox_labels=ds.num_bbox_labels,
             label_shape=np.array(ds.label_shape, dtype=np.int64), has_labels=ds.has_labels,
             dims=np.array([ds.num_assets, ds.num_channels, ds.height, ds.width, ds.background_size_for_training, ds.label_dim]))
    for i in range(len(ds)):
        s, label = ds[i]
        d[f's{i}/label'] = label
        for k in ('bboxes', 'labels', 'mask', 'background', 'background_orig', 'patches_orig', 'patch_masks'):
            d[f's{i}/{k}'] = s[k]
        d[f's{i}/texts'] = np.array(s['texts'])
        d[f's{i}/meta'] = np.array([s['name'], str(s['W_page']), str(s['H_page'])])
        d[f's{i}/patches_sub4'] = s['patches'][:, :, ::4, ::4]
        d[f's{i}/patches_sum'] = s['patches'].astype(np.float64).sum(axis=(1, 2, 3))
        d[f's{i}/patches_abs_sum'] = np.abs(s['patches'].astype(np.float64)).sum(axis=(1, 2, 3))
    ds2 = LayoutDataset(path=zpath, background_size=BG_SIZE, use_labels=False, max_size=2, random_seed=3)
    d['max_size2_raw_idx'] = ds2._raw_idx
    path = os.path.join(OUT, 'dataset.npz')
    np.savez_compressed(path, **d)
    print('wrote', path, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    main()
