"""the zip + `non_image.json` reader (layoutdetr_amd/training/dataset_layoutganpp.py::LayoutDataset).

tests/golden/dataset_tiny.zip is a 3-sample archive in the reference's on-disk format and tests/golden/dataset.npz is what the REFERENCE's
own `training.dataset_layoutganpp.LayoutDataset.__getitem__` returned for it (oracle/gen_dataset_golden.py made both).
CPU tests: mode='reference' reproduces every key of the reference's items; mode='device' never opens a patch PNG and ships uint8 pages.
GPU tests: the device-side resize + normalise of those pages equals the reference's host-side 'background' bit for bit, and
`training_loop()` trains from the archive through the DataLoader."""
import os
import pickle
import zipfile

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ZIP = os.path.join(ROOT, 'tests', 'golden', 'dataset_tiny.zip')


@pytest.fixture(scope='module')
def gold():
    return np.load(os.path.join(ROOT, 'tests', 'golden', 'dataset.npz'), allow_pickle=False)


def _ds(**kw):
    from layoutdetr_amd.training.dataset_layoutganpp import LayoutDataset
    return LayoutDataset(path=ZIP, background_size=32, use_labels=False, max_size=None, xflip=False, **kw)


def test_reference_mode_reproduces_the_reference_items(gold):
    """Every key of the reference's item (dataset_layoutganpp.py:267-342), bit for bit: same Pillow, same float32 operation order."""
    ds = _ds(mode='reference')
    assert len(ds) == int(gold['n']) == 3 and ds.name == str(gold['name'])
    assert ds.patch_shape == list(gold['patch_shape']) and ds.num_bbox_labels == int(gold['num_bbox_labels'])
    assert [ds.num_assets, ds.num_channels, ds.height, ds.width, ds.background_size_for_training, ds.label_dim] == list(gold['dims'])
    assert ds.label_shape == list(gold['label_shape']) and ds.has_labels == bool(gold['has_labels'])
    for i in range(3):
        s, label = ds[i]
        assert label.dtype == np.float32 and np.array_equal(label, gold[f's{i}/label'])
        for k in ('bboxes', 'labels', 'mask', 'background', 'background_orig', 'patches_orig', 'patch_masks'):
            want = gold[f's{i}/{k}']
            assert s[k].dtype == want.dtype and s[k].shape == want.shape, (i, k, s[k].dtype, want.dtype, s[k].shape, want.shape)
            assert np.array_equal(s[k], want), (i, k)
        assert s['texts'] == list(gold[f's{i}/texts']) and len(s['texts']) == 9
        assert [s['name'], str(s['W_page']), str(s['H_page'])] == list(gold[f's{i}/meta'])
        assert s['patches'].shape == (9, 3, 256, 256) and s['patches'].dtype == np.float32
        assert np.array_equal(s['patches'][:, :, ::4, ::4], gold[f's{i}/patches_sub4'])
        assert np.array_equal(s['patches'].astype(np.float64).sum(axis=(1, 2, 3)), gold[f's{i}/patches_sum'])
        assert np.array_equal(np.abs(s['patches'].astype(np.float64)).sum(axis=(1, 2, 3)), gold[f's{i}/patches_abs_sum'])
    from layoutdetr_amd.training.dataset_layoutganpp import LayoutDataset
    assert np.array_equal(LayoutDataset(path=ZIP, background_size=32, max_size=2, random_seed=3)._raw_idx, gold['max_size2_raw_idx'])


def test_device_mode_never_opens_a_patch_and_ships_uint8_pages(gold, monkeypatch):
    ds = _ds()
    assert ds.mode == 'device' and ds.patch_shape == list(gold['patch_shape'])      # learnt from a PNG header, no pixel decoded
    opened = []
    real_open = ds._open_file
    monkeypatch.setattr(ds, '_open_file', lambda fname: (opened.append(fname), real_open(fname))[1])
    items = [ds[i] for i in range(3)]
    assert opened and all(f.endswith('_background_orig.png') for f in opened), opened
    with zipfile.ZipFile(ZIP) as z:
        import PIL.Image
        for i, (s, label) in enumerate(items):
            for k in ('bboxes', 'labels', 'mask'):
                assert np.array_equal(s[k], gold[f's{i}/{k}']) and s[k].dtype == gold[f's{i}/{k}'].dtype
            assert s['texts'] == list(gold[f's{i}/texts'])
            assert s['background'].dtype == np.uint8 and s['background'].shape == (56, 80, 3)
            assert np.array_equal(s['background'], np.array(PIL.Image.open(z.open(opened[i]))))
            assert s['patches'].shape == (9, 3, 256, 256) and s['patches'].strides == (0, 0, 0, 0)
            assert 'patches_orig' not in s and 'patch_masks' not in s and 'background_orig' not in s
    # the collate keeps pages uint8 and the placeholder 0-stride; texts arrive as the default collate would deliver them
    batch, labels = ds.collate(items)
    assert batch['background'].dtype == torch.uint8 and tuple(batch['background'].shape) == (3, 56, 80, 3)
    assert tuple(batch['patches'].shape) == (3, 9, 3, 256, 256) and batch['patches'].stride(0) == 0 and batch['patches'].untyped_storage().nbytes() <= 64
    assert batch['bboxes'].dtype == torch.float32 and batch['labels'].dtype == torch.int64 and batch['mask'].dtype == torch.bool
    assert tuple(labels.shape) == (3, 0)
    assert list(map(list, zip(*batch['texts']))) == [s['texts'] for s, _ in items]
    ref_batch = torch.utils.data.default_collate([({k: v for k, v in s.items() if k in ('bboxes', 'labels', 'mask', 'texts')}, l) for s, l in items])[0]
    assert [tuple(t) for t in ref_batch['texts']] == batch['texts'] and torch.equal(ref_batch['bboxes'], batch['bboxes'])
    # pages of different sizes stay a list
    odd = [(dict(items[0][0], background=items[0][0]['background'][:40]), items[0][1]), items[1]]
    assert isinstance(ds.collate(odd)[0]['background'], list)


def test_dataset_survives_pickling_and_dataloader_workers():
    ds = _ds()
    _ = ds[0]
    ds2 = pickle.loads(pickle.dumps(ds))
    assert ds2._zipfile is None and np.array_equal(ds2[1][0]['background'], ds[1][0]['background'])
    dl = torch.utils.data.DataLoader(ds, batch_size=3, collate_fn=ds.collate, num_workers=2)
    batch, labels = next(iter(dl))
    assert tuple(batch['background'].shape) == (3, 56, 80, 3) and batch['mask'].sum().item() == 9 + 4 + 1
    ds.close(); ds2.close()


def test_malformed_archives_raise(tmp_path):
    from layoutdetr_amd.training.dataset_layoutganpp import LayoutDataset, to_dense_batch
    with pytest.raises(IOError, match='zip'):
        LayoutDataset(path=str(tmp_path))
    p = tmp_path / 'a' / 'b' / 'empty.zip'
    p.parent.mkdir(parents=True)
    with zipfile.ZipFile(p, 'w') as z:
        z.writestr('readme.txt', 'x')
    with pytest.raises(IOError, match='non_image.json'):
        LayoutDataset(path=str(p))
    with pytest.raises(ValueError, match='at most 9'):
        to_dense_batch(np.zeros((10, 4)))
    d, m = to_dense_batch(np.zeros((0, 4)))
    assert d.shape == (9, 4) and not m.any()
    with pytest.raises(ValueError, match='mode'):
        LayoutDataset(path=ZIP, mode='host')


@pytest.mark.gpu
def test_device_backgrounds_equal_the_reference_host_path_bit_for_bit(dev, gold):
    from layoutdetr_amd.training.dataset_layoutganpp import batch_backgrounds_to_device, patch_placeholder_to_device
    ds = _ds()
    batch, _ = ds.collate([ds[i] for i in range(3)])
    bg = batch_backgrounds_to_device(batch['background'], ds.background_size_for_training, dev)
    want = torch.from_numpy(np.stack([gold[f's{i}/background'] for i in range(3)]))
    assert bg.dtype == torch.float32 and tuple(bg.shape) == (3, 3, 32, 32)
    assert torch.equal(bg.cpu(), want), float((bg.cpu() - want).abs().max())
    # a ragged list of pages takes one resize per page size
    pages = [batch['background'][0], batch['background'][1][:40].contiguous(), batch['background'][2]]
    bg2 = batch_backgrounds_to_device(pages, 32, dev)
    assert torch.equal(bg2[0], bg[0]) and torch.equal(bg2[2], bg[2]) and torch.isfinite(bg2[1]).all() and not torch.equal(bg2[1], bg[1])
    pp = patch_placeholder_to_device(batch['patches'], dev)
    assert pp.device.type == 'cuda' and tuple(pp.shape) == (3, 9, 3, 256, 256) and pp.stride(0) == 0


@pytest.mark.gpu
def test_training_loop_trains_from_the_archive(dev, tmp_path):
    """train.py's own dataset kwargs (train.py:107: class_name 'training.dataset_layoutganpp.LayoutDataset', path, use_labels, max_size,
    xflip, background_size) through the reference's module names; strings from non_image.json through the host tokenizer."""
    import importlib
    from layoutdetr_amd import dropin
    dropin.install()
    try:
        tl = importlib.import_module('training.training_loop')
        words = set()
        ds = _ds()
        for s in ds._samples:
            for t in s[1]['texts']:
                words.update(t.replace('%', ' % ').replace('!', ' !').split())
        vf = tmp_path / 'vocab.txt'
        vf.write_text('\n'.join(['[PAD]', '[unused0]', '[UNK]', '[CLS]', '[SEP]', '[MASK]'] + sorted(words)) + '\n')
        net = dict(bert_f_dim=768, bert_num_heads=4, bert_num_encoder_layers=2, bert_num_decoder_layers=2, im_f_dim=512, text_mode='encoder', tokenizer_vocab=str(vf))
        out = tl.training_loop(
            run_dir=str(tmp_path), training_set_kwargs=dict(class_name='training.dataset_layoutganpp.LayoutDataset', path=ZIP, use_labels=False, max_size=3, xflip=False, background_size=64),
            data_loader_kwargs=dict(num_workers=1, prefetch_factor=2), random_seed=0, num_gpus=1, rank=0, batch_size=2, batch_gpu=2,
            G_kwargs=dict(class_name='training.networks_detr.Generator', z_dim=4, **net), D_kwargs=dict(class_name='training.networks_detr.Discriminator', **net),
            G_opt_kwargs=dict(class_name='torch.optim.Adam', betas=[0, 0.99], eps=1e-8, lr=1e-5), D_opt_kwargs=dict(class_name='torch.optim.Adam', betas=[0, 0.99], eps=1e-8, lr=1e-5),
            loss_kwargs=dict(class_name='training.loss.StyleGAN2Loss', r1_gamma=0.0, pl_weight=0.0), G_reg_interval=4, D_reg_interval=16,
            ema_kimg=2 * 10 / 32, total_kimg=0.006, kimg_per_tick=0.002, network_snapshot_ticks=None)
    finally:
        dropin.uninstall()
    assert out['stats']['cur_nimg'] == 6
    for k in ('Loss/scores/fake', 'Loss/scores/real', 'Loss/G/loss_Ggen_bbox_rec', 'Loss/D/loss_Dreal_bg_rec'):
        assert k in out['stats'] and np.isfinite(out['stats'][k]), k
    assert all(torch.isfinite(p).all() for p in out['G'].parameters())
