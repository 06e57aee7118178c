"""Drop-in boundary on the GPU: the seam-2 wrappers with the reference's signatures (conv2d_resample, conv2d_gradfix),
the pybind-shaped plugin bindings a maintainer would add (layoutdetr_amd.dropin), and the seam-1 `training_loop(**c)` entry driven
the way the reference's train.py drives it (class names, dataset object, strings as bbox_text)."""
import os

import numpy as np
import pytest
import torch

from oracle import ops_ref

pytestmark = pytest.mark.gpu
G_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load(name):
    d = np.load(os.path.join(G_DIR, name + '.npz'), allow_pickle=False)
    return {k: torch.from_numpy(d[k]) for k in d.files if d[k].ndim > 0 or d[k].dtype.kind in 'fiub'}


def close(a, b, tol, what=''):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    e = ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()
    assert e <= tol, f'{what}: rel err {e:.3e}'


def test_conv2d_resample_matches_reference_golden_and_gradients(dev):
    """All four branch families against vectors captured from the reference's conv2d_resample (ops.npz cr_*), gradients against
    autograd through the golden-pinned oracle."""
    from layoutdetr_amd.torch_utils.ops import conv2d_resample
    d = load('ops')
    f = d['f']
    cases = [('cr_up2', 'cr_w3', dict(f=f, up=2, padding=1, flip_weight=False)), ('cr_up1', 'cr_w3', dict(f=f, up=1, padding=1, flip_weight=True)),
             ('cr_1x1', 'cr_w1', dict(f=None, up=1, padding=0, flip_weight=True)), ('cr_down2', 'cr_w3', dict(f=f, down=2, padding=1))]
    for key, wk, kw in cases:
        x = d['cr_x'].to(dev).requires_grad_(True); w = d[wk].to(dev).requires_grad_(True)
        kg = dict(kw); kg['f'] = None if kw['f'] is None else kw['f'].to(dev)
        y = conv2d_resample.conv2d_resample(x, w, **kg)
        close(y, d[key], 2e-5, key)
        g = torch.randn(y.shape, generator=torch.Generator().manual_seed(3))
        y.backward(g.to(dev))
        xr = d['cr_x'].clone().requires_grad_(True); wr = d[wk].clone().requires_grad_(True)
        yr = ops_ref.conv2d_resample(xr, wr, **{k: v for k, v in kw.items() if k != 'flip_filter'})
        yr.backward(g)
        close(x.grad, xr.grad, 3e-5, key + ' dx'); close(w.grad, wr.grad, 3e-5, key + ' dw')


def test_conv2d_gradfix_surface(dev):
    from layoutdetr_amd.torch_utils.ops import conv2d_gradfix
    import torch.nn.functional as F
    torch.manual_seed(1)
    x = torch.randn(2, 8, 9, 7); w = torch.randn(12, 8, 3, 3) * 0.2; b = torch.randn(12); wt = torch.randn(8, 6, 3, 3) * 0.2
    for fn, ref, wgt, kw in ((conv2d_gradfix.conv2d, F.conv2d, w, dict(stride=2, padding=1)), (conv2d_gradfix.conv2d, F.conv2d, w, dict(padding=[1, 1])),
                             (conv2d_gradfix.conv_transpose2d, F.conv_transpose2d, wt, dict(stride=2, padding=1))):
        bias = b[:wgt.shape[0] if fn is conv2d_gradfix.conv2d else wgt.shape[1]]
        xr = x.clone().requires_grad_(True); wr = wgt.clone().requires_grad_(True); br = bias.clone().requires_grad_(True)
        yr = ref(xr, wr, br, **kw); g = torch.randn_like(yr); yr.backward(g)
        xg = x.to(dev).requires_grad_(True); wg = wgt.to(dev).requires_grad_(True); bg = bias.to(dev).requires_grad_(True)
        y = fn(xg, wg, bg, **kw); y.backward(g.to(dev))
        close(y, yr, 2e-5, 'y'); close(xg.grad, xr.grad, 3e-5, 'dx'); close(wg.grad, wr.grad, 3e-5, 'dw'); close(bg.grad, br.grad, 3e-5, 'db')
    assert conv2d_gradfix.enabled is True
    xg = x.to(dev).requires_grad_(True); wg = w.to(dev).requires_grad_(True)
    with conv2d_gradfix.no_weight_gradients():
        assert conv2d_gradfix.weight_gradients_disabled
        conv2d_gradfix.conv2d(xg, wg, padding=1).sum().backward()
    assert wg.grad is None and xg.grad is not None and not conv2d_gradfix.weight_gradients_disabled
    with pytest.raises(NotImplementedError):
        conv2d_gradfix.conv2d(xg, wg, groups=2)


def test_plugin_bindings_with_the_pybind_argument_lists(dev):
    """dropin.bias_act_plugin.bias_act(x, b, xref, yref, dy, grad, dim, act, alpha, gain, clamp) and
    dropin.upfirdn2d_plugin.upfirdn2d(x, f, upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip, gain): the calls the reference's
    torch_utils/ops wrappers make (bias_act.py:140,178,199; upfirdn2d.py:238), empty tensor = absent, RuntimeError on bad arguments."""
    from layoutdetr_amd import dropin
    d = load('ops')
    null = torch.empty([0], device=dev)
    x = d['ba_x'].to(dev); b = d['ba_b'].to(dev)
    y = dropin.bias_act_plugin.bias_act(x, b, null, null, null, 0, 1, 3, 0.2, 2 ** 0.5, -1.0)         # lrelu = cuda_idx 3
    close(y, d['ba_lrelu_y'], 2e-6, 'bias_act fwd')
    dy = torch.ones_like(y) * 0.5 + y * 0.1
    dx = dropin.bias_act_plugin.bias_act(dy, b, x, y, null, 1, 1, 3, 0.2, 2 ** 0.5, -1.0)
    close(dx, d['ba_lrelu_dx'], 2e-6, 'bias_act grad=1')
    xcl = x.contiguous(memory_format=torch.channels_last)
    ycl = dropin.bias_act_plugin.bias_act(xcl, b, null, null, null, 0, 1, 3, 0.2, 2 ** 0.5, -1.0)
    assert ycl.is_contiguous(memory_format=torch.channels_last); close(ycl, d['ba_lrelu_y'], 2e-6, 'channels_last')
    with pytest.raises(RuntimeError, match='wrong number of elements'):
        dropin.bias_act_plugin.bias_act(x, b[:3], null, null, null, 0, 1, 3, 0.2, 1.0, -1.0)
    with pytest.raises(RuntimeError, match='same layout'):
        dropin.bias_act_plugin.bias_act(x, b, xcl, null, null, 1, 1, 3, 0.2, 1.0, -1.0)
    fa = d['fa'].to(dev); xu = d['up_x'].to(dev)
    # case 1 of gen_ops: up=2, pad [2,1,2,1], gain 4 -> the wrapper passes gain * up^0... exactly: upfirdn2d.py hands `gain` through
    y = dropin.upfirdn2d_plugin.upfirdn2d(xu, fa, 2, 2, 1, 1, 2, 1, 2, 1, False, 4.0)
    close(y, d['up1_y'], 3e-6, 'upfirdn2d plugin')
    y = dropin.upfirdn2d_plugin.upfirdn2d(xu.contiguous(memory_format=torch.channels_last), fa, 3, 3, 2, 2, -1, 4, 2, 0, True, 1.5)
    assert y.is_contiguous(memory_format=torch.channels_last); close(y, d['up4_y'], 3e-6, 'upfirdn2d plugin flip / crop')
    with pytest.raises(RuntimeError, match='rank 2'):
        dropin.upfirdn2d_plugin.upfirdn2d(xu, fa[0], 1, 1, 1, 1, 0, 0, 0, 0, False, 1.0)


class SyntheticLayouts(torch.utils.data.Dataset):
    """Stands for training.dataset_layoutganpp.LayoutDataset: the attributes training_loop reads (:127-132) and items shaped as its
    __getitem__ returns them (dict of bboxes / labels / texts / patches / mask / background, label)."""
    num_bbox_labels, num_channels, height, width, label_dim = 8, 3, 64, 64, 0

    def __init__(self, n=8, background_size_for_training=64):
        self.n, self.background_size_for_training = n, background_size_for_training
        self.words = ['sale', 'shop now', 'up to 50% off', 'new', 'free shipping', 'ok', 'limited time only!', 'x', 'sign up']

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        i = int(i)
        g = torch.Generator().manual_seed(i)
        bb = torch.cat([torch.rand(9, 2, generator=g) * 0.6 + 0.2, torch.rand(9, 2, generator=g) * 0.35 + 0.05], -1)
        mask = torch.ones(9, dtype=torch.bool); mask[5 + i % 4:] = False
        return dict(bboxes=bb.numpy(), labels=torch.randint(0, 8, (9,), generator=g).numpy(), texts=[self.words[(i + j) % 9] for j in range(9)],
                    patches=np.zeros((9, 3, 4, 4), np.float32), mask=mask.numpy(),
                    background=torch.randn(3, 64, 64, generator=g).numpy()), np.zeros((0,), np.float32)


@pytest.fixture
def reference_names():
    """dropin.install() for the duration of one test (it also switches the constructors to the reference's defaults: process-wide)."""
    from layoutdetr_amd import dropin
    dropin.install()
    try:
        yield dropin
    finally:
        dropin.uninstall()


def _write_vocab(tmp_path):
    vocab = ['[PAD]', '[unused0]', '[UNK]', '[CLS]', '[SEP]', '[MASK]'] + sorted({w for s in SyntheticLayouts().words for w in s.replace('%', ' % ').replace('!', ' !').split()})
    vf = tmp_path / 'vocab.txt'
    vf.write_text('\n'.join(vocab) + '\n')
    return vf


def test_training_loop_with_train_py_own_kwargs(dev, tmp_path, reference_names, monkeypatch):
    """What `python -m layoutdetr_amd.dropin train.py ...` does: G_kwargs / D_kwargs EXACTLY as train.py builds them (class_name + the
    option-derived fields of train.py:250-261 — no text_mode, no tokenizer_vocab), strings from the loader.  Through the reference's
    module names the constructors default to what the reference always builds (tokenizer from LDETR_BERT_VOCAB, frozen text encoder, LM
    text decoder); a missing vocabulary fails at construction with a message, not at the first iteration.  Also exercised: resume_pkl
    (a snapshot of a previous run), ema_rampup=None, the final network snapshot, stats.jsonl."""
    import importlib
    import json
    import pickle
    tl = importlib.import_module('training.training_loop')
    nd = importlib.import_module('training.networks_detr')
    common = dict(num_bbox_labels=8, img_channels=3, img_height=64, img_width=64, background_size=64, c_dim=0)
    train_py = dict(f_dim=256, num_heads=4, num_layers=8, bert_f_dim=768, bert_num_heads=4, bert_num_encoder_layers=2, bert_num_decoder_layers=2, im_f_dim=512)
    monkeypatch.delenv('LDETR_BERT_VOCAB', raising=False)
    with pytest.raises(RuntimeError, match='LDETR_BERT_VOCAB'):
        nd.Generator(z_dim=4, **train_py, **common)
    monkeypatch.setenv('LDETR_BERT_VOCAB', str(_write_vocab(tmp_path)))
    G = nd.Generator(z_dim=4, **train_py, **common)
    assert G.text_mode == 'encoder+lm' and G.text_decoder is not None and G.tokenizer is not None
    del G
    run1, run2 = tmp_path / 'run1', tmp_path / 'run2'
    run1.mkdir(); run2.mkdir()
    kw = dict(training_set_kwargs=dict(class_name='test_boundary_gpu.SyntheticLayouts', n=8), data_loader_kwargs=dict(num_workers=0), random_seed=0,
              num_gpus=1, rank=0, batch_size=4, batch_gpu=4,
              G_kwargs=dict(class_name='training.networks_detr.Generator', z_dim=4, **train_py), D_kwargs=dict(class_name='training.networks_detr.Discriminator', **train_py),
              G_opt_kwargs=dict(class_name='torch.optim.Adam', betas=[0, 0.99], eps=1e-8, lr=1e-5), D_opt_kwargs=dict(class_name='torch.optim.Adam', betas=[0, 0.99], eps=1e-8, lr=1e-5),
              loss_kwargs=dict(class_name='training.loss.StyleGAN2Loss', r1_gamma=0.0, pl_weight=0.0), G_reg_interval=4, D_reg_interval=16,
              ema_kimg=4 * 10 / 32, total_kimg=0.008, kimg_per_tick=0.004, network_snapshot_ticks=50)
    out = tl.training_loop(run_dir=str(run1), **kw)
    assert out['stats']['cur_nimg'] == 8 and out['snapshot_pkl'] and os.path.exists(out['snapshot_pkl'])
    assert np.isfinite(out['stats']['Loss/G/loss_Ggen_text_rec']) and out['stats']['Loss/G/loss_Ggen_text_rec'] > 0, 'the LM text decoder did not run'
    lines = [json.loads(l) for l in open(run1 / 'stats.jsonl')]
    assert len(lines) >= 2 and lines[-1]['Loss/scores/fake']['num'] > 0 and 'std' in lines[-1]['Loss/scores/real']
    with open(out['snapshot_pkl'], 'rb') as f:
        snap = pickle.load(f)
    assert set(snap) >= {'G', 'D', 'G_ema', 'training_set_kwargs'} and not any(p.requires_grad for p in snap['G_ema'].parameters())
    assert all(p.device.type == 'cpu' for p in snap['D'].parameters())
    for (n, a), (_, b) in zip(snap['G'].named_parameters(), out['G'].named_parameters()):
        assert torch.equal(a, b.detach().cpu()), n
    # resume: the second run starts from the first run's weights (rank 0 loads, parameters and buffers copied by name)
    out2 = tl.training_loop(run_dir=str(run2), resume_pkl=out['snapshot_pkl'], resume_kimg=0, ema_rampup=None, **dict(kw, total_kimg=0.004))
    w1 = dict(snap['D'].named_parameters())['enc_fc_in.layers.0.weight']
    w2 = dict(out2['D'].named_parameters())['enc_fc_in.layers.0.weight'].detach().cpu()
    assert (w1 - w2).abs().max() < 1e-3 and not torch.equal(w1, w2), 'run 2 did not start from the snapshot (or did not train)'
    # ema_rampup=None: beta = 0.5 ** (batch / (ema_kimg * 1000)) = 0.5 ** (4 / 1250), so G_ema stays within (1 - beta) of the snapshot's G_ema
    e1 = dict(snap['G_ema'].named_parameters())['fc_in.layers.0.weight']; e2 = dict(out2['G_ema'].named_parameters())['fc_in.layers.0.weight'].detach().cpu()
    g2 = dict(out2['G'].named_parameters())['fc_in.layers.0.weight'].detach().cpu()
    beta = 0.5 ** (4 / 1250.0)
    assert torch.allclose(e2, g2.lerp(e1, beta), atol=1e-6), 'ema_rampup=None was not honoured'


def test_training_loop_entry_runs_like_train_py_drives_it(dev, tmp_path, reference_names):
    """training_loop(**c) with the reference's keyword arguments (train.py:47,197-283): networks and loss by class name THROUGH THE
    REFERENCE'S MODULE NAMES (dropin.install()), a dataset object, strings as bbox_text (host WordPiece tokenizer from a local vocab)."""
    import importlib
    tl = importlib.import_module('training.training_loop')
    vf = _write_vocab(tmp_path)
    net = dict(bert_f_dim=768, bert_num_heads=4, bert_num_encoder_layers=2, bert_num_decoder_layers=2, im_f_dim=512, text_mode='encoder', tokenizer_vocab=str(vf))
    seen = []
    out = tl.training_loop(
        run_dir=str(tmp_path), training_set_kwargs=dict(class_name='test_boundary_gpu.SyntheticLayouts', n=8),
        data_loader_kwargs=dict(num_workers=0), random_seed=0, num_gpus=1, rank=0, batch_size=4, batch_gpu=4,
        G_kwargs=dict(class_name='training.networks_detr.Generator', z_dim=4, **net), D_kwargs=dict(class_name='training.networks_detr.Discriminator', **net),
        G_opt_kwargs=dict(class_name='torch.optim.Adam', betas=[0, 0.99], eps=1e-8, lr=1e-5), D_opt_kwargs=dict(class_name='torch.optim.Adam', betas=[0, 0.99], eps=1e-8, lr=1e-5),
        loss_kwargs=dict(class_name='training.loss.StyleGAN2Loss', r1_gamma=0.0, pl_weight=0.0, Ggen_bbox_rec_weight=100.0), G_reg_interval=4, D_reg_interval=16,
        ema_kimg=4 * 10 / 32, total_kimg=0.012, kimg_per_tick=0.004, progress_fn=lambda cur, tot: seen.append(cur))
    assert out['stats']['cur_nimg'] == 12 and len(seen) >= 3
    assert type(out['G']).__module__ == 'layoutdetr_amd.training.networks_detr'
    for k in ('Loss/scores/fake', 'Loss/scores/real', 'Loss/G/loss_Ggen_bbox_rec', 'Loss/D/loss_Dreal_bg_rec'):
        assert k in out['stats'] and np.isfinite(out['stats'][k]), k
    assert all(torch.isfinite(p).all() for p in out['G'].parameters())
    moved = sum(float((a - b).abs().max()) > 0 for a, b in zip(out['G'].parameters(), out['G_ema'].parameters()))
    assert moved > 100, 'G was not updated / EMA not tracking'


def test_bench_two_ranks_on_one_gpu_prints_the_scaling_schema(dev):
    """`python bench.py --gpus 2` with both ranks on cuda:0 over gloo (LDETR_BENCH_SHARE_GPU=1): the rank logic, the staged graphs, the
    overlapped exchange plumbing and -- what this test is for -- the ONE JSON line the driver's scaling sweep parses, with every diagnostic
    field the first real multi-GPU run is read through.  No xGMI is involved; values are not asserted, the schema is."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LDETR_BENCH_SHARE_GPU='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--no-roofline'], env=env, cwd=root,
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, f'expected exactly one JSON line on stdout, got {len(lines)}'
    d = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data', 'config',
              'rccl_ranks', 'diagnostics', 'weak_scaling', 'single_gpu_reference', 'strong_scaling_ceiling'):
        assert k in d, f'missing field {k}'
    assert d['n_gpus'] == 2 and d['rccl_ranks'] == 2 and d['steps'] == 3 and d['warmup'] == 1 and d['scaling'] == 'strong' and d['higher_is_better'] is True
    assert d['unit'] == 'images/s' and d['value'] > 0 and d['ms_per_step'] > 0 and d['shared_single_gpu_gloo'] is True
    assert d['config']['global_batch'] == 16 and d['config']['per_gpu_batch'] == 8 and d['config']['parallelism'] == 'dp2' and 'workload' in d['config']
    assert d['config']['allreduce_overlapped_with_backward'] is True
    diag = d['diagnostics']
    assert len(diag['rank_ms_per_step']['per_rank']) == 2 and set(diag['comm_exposed_ms']) >= {'Gmain', 'Dmain', 'per_step_total_max'} and diag['allreduce_mb_per_step'] > 0
    w = d['weak_scaling']
    assert w['global_batch'] == 32 and w['per_gpu_batch'] == 16 and w['value'] > 0 and 'comm_exposed_ms' in w['diagnostics']
    assert d['single_gpu_reference']['per_gpu_batch'] == 16 and d['single_gpu_reference']['value_per_gpu'] > 0
    c = d['strong_scaling_ceiling']
    assert c['n_gpus'] == 2 and c['per_gpu_batch_share'] == 8 and 1.0 < c['value'] <= 2.05
