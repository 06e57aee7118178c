"""CPU tests: pin the oracle (oracle/*.py) to golden vectors captured from the reference itself
(tests/golden/*.npz, produced by oracle/gen_golden.py importing /root/reference).  fp32, tolerance 2e-5
relative unless stated (summation-order differences only); index outputs bit-exact."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

from oracle import detr_ref, losses_ref, ops_ref, stylegan2_ref

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(name):
    d = np.load(os.path.join(G, name + '.npz'), allow_pickle=False)
    return {k: torch.from_numpy(d[k]) if d[k].dtype.kind in 'fbiu' and d[k].ndim > 0 else d[k] for k in d.files}


def sd_of(d, prefix='sd/'):
    return {k[len(prefix):]: v for k, v in d.items() if k.startswith(prefix)}


def close(a, b, tol=2e-5):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    err = (a - b).abs().max().item() / (b.abs().max().item() + 1e-12)
    assert err <= tol, f'rel err {err:.3e}'


def test_ops_bias_act():
    d = load('ops')
    for act in ops_ref.ACT_DEFAULTS:
        x = d['ba_x'].clone().requires_grad_(True); b = d['ba_b'].clone().requires_grad_(True)
        y = ops_ref.bias_act(x, b, act=act)
        y.backward(torch.ones_like(y) * 0.5 + y.detach() * 0.1)
        close(y, d[f'ba_{act}_y']); close(x.grad, d[f'ba_{act}_dx']); close(b.grad, d[f'ba_{act}_db'])
        close(ops_ref.bias_act(d['ba_x'], d['ba_b'], act=act, alpha=0.1, gain=0.7, clamp=0.9), d[f'ba_{act}_yc'])


def test_ops_upfirdn2d_and_resample():
    d = load('ops')
    cases = [dict(up=1, down=1, padding=[1, 1, 1, 1], gain=4), dict(up=2, down=1, padding=[2, 1, 2, 1], gain=4),
             dict(up=1, down=2, padding=[1, 1, 1, 1], gain=1), dict(up=[2, 1], down=[1, 3], padding=[0, 2, -1, 3], gain=0.5),
             dict(up=3, down=2, padding=[-1, 4, 2, 0], gain=1.5, flip_filter=True)]
    close(ops_ref.setup_filter([1, 3, 3, 1]), d['f'], 1e-7)
    for i, c in enumerate(cases):
        x = d['up_x'].clone().requires_grad_(True)
        y = ops_ref.upfirdn2d(x, d['fa'], **c)
        y.backward(torch.ones_like(y) + 0.1 * y.detach())
        close(y, d[f'up{i}_y']); close(x.grad, d[f'up{i}_dx'])
    close(ops_ref.upsample2d(d['up_x'], d['f']), d['up2d_y'])
    f = d['f']
    close(ops_ref.conv2d_resample(d['cr_x'], d['cr_w3'], f=f, up=2, padding=1, flip_weight=False), d['cr_up2'])
    close(ops_ref.conv2d_resample(d['cr_x'], d['cr_w3'], f=f, up=1, padding=1, flip_weight=True), d['cr_up1'])
    close(ops_ref.conv2d_resample(d['cr_x'], d['cr_w1'], f=None, up=1, padding=0), d['cr_1x1'])
    close(ops_ref.conv2d_resample(d['cr_x'], d['cr_w3'], f=f, down=2, padding=1), d['cr_down2'])


def test_ops_modulated_conv():
    d = load('ops')
    for nm, kw in [('up2', dict(up=2, padding=1, resample_filter=d['f'], flip_weight=False)), ('up1', dict(up=1, padding=1, flip_weight=True))]:
        x = d['cr_x'].clone().requires_grad_(True); w = d['cr_w3'].clone().requires_grad_(True); s = d['mc_s'].clone().requires_grad_(True)
        y = ops_ref.modulated_conv2d(x, w, s, **kw)
        y.backward(torch.ones_like(y) + 0.1 * y.detach())
        close(y, d[f'mc_{nm}_y']); close(x.grad, d[f'mc_{nm}_dx']); close(w.grad, d[f'mc_{nm}_dw']); close(s.grad, d[f'mc_{nm}_ds'])
    close(ops_ref.modulated_conv2d(d['cr_x'], d['cr_w1'], d['mc_s'], demodulate=False), d['mc_rgb_y'])


@pytest.mark.parametrize('name,with_token', [('transformer', False), ('transformer_token', True)])
def test_detr_transformer(name, with_token):
    d = load(name)
    sd = {k: v.clone().requires_grad_(v.dtype.is_floating_point) for k, v in sd_of(d).items()}
    src = d['src'].clone().requires_grad_(True); tgt = d['tgt'].clone().requires_grad_(True)
    hs, mem = detr_ref.transformer(sd, src, d['mask'], d['pos'], tgt, d['kpm'], nhead=2, with_token=with_token)
    close(hs, d['hs']); close(mem, d['mem'])
    ((hs * d['g_hs']).sum() + (mem * d['g_mem']).sum()).backward()
    close(src.grad, d['d_src'], 5e-5); close(tgt.grad, d['d_tgt'], 5e-5)
    for k, g in sd_of(d, 'grad/').items():
        close(sd[k].grad, g, 1e-4)


def test_torch_encoder_branch():
    d = load('transformer_layoutganpp')
    sd = {k: v.clone().requires_grad_(v.dtype.is_floating_point) for k, v in sd_of(d).items()}
    x = d['x'].clone().requires_grad_(True)
    y = detr_ref.token_encoder_layoutganpp(sd, '', x, d['kpm'], nhead=2)
    close(y, d['y'])
    (y * d['g']).sum().backward()
    close(x.grad, d['d_x'], 5e-5)
    for k, g in sd_of(d, 'grad/').items():
        close(sd[k].grad, g, 1e-4)


def test_position_encoding_and_frozen_bn():
    d = load('pos_encoding')
    close(detr_ref.position_embedding_sine(d['mask']), d['pos'], 1e-6)
    d = load('frozen_bn')
    close(detr_ref.frozen_bn(d, '', d['x']), d['y'], 1e-6)


def test_layout_losses():
    d = load('losses')
    for nm, fn in [('overlap', losses_ref.compute_overlap), ('alignment', losses_ref.compute_alignment)]:
        b = d['bbox'].clone().requires_grad_(True)
        v = fn(b, d['mask']); v.sum().backward()
        close(v, d[nm]); close(b.grad, d['d_' + nm], 5e-5)
    b = d['bbox'].clone().requires_grad_(True)
    v = losses_ref.generalized_iou_loss(b[d['mask']], d['real'][d['mask']]); v.backward()
    close(v, d['giou']); close(b.grad, d['d_giou'], 5e-5)


def test_stylegan2_decoder():
    d = load('decoder')
    sd = {k: v.clone().requires_grad_(v.dtype.is_floating_point) for k, v in sd_of(d).items()}
    z = d['z'].clone().requires_grad_(True)
    img = stylegan2_ref.decoder(sd, '', z, 16)
    close(img, d['img'])
    (img * d['g']).sum().backward()
    close(z.grad, d['d_z'], 1e-4)
    for k, g in sd_of(d, 'grad/').items():
        close(sd[k].grad, g, 2e-4)


def test_dp_postprocess():
    d = load('dp_step')
    flat = torch.cat([d[f'g{i}'].flatten() for i in range(3)])
    for W in (1, 2, 8):
        out = losses_ref.dp_postprocess(flat * W, W)
        assert torch.equal(out, d[f'out_w{W}'])


def _lsap_lib():
    so = os.path.join(ROOT, 'oracle', '_build', 'liblsap_oracle.so')
    if not os.path.exists(so):
        os.makedirs(os.path.dirname(so), exist_ok=True)
        subprocess.check_call(['gcc', '-O2', '-shared', '-fPIC', '-o', so, os.path.join(ROOT, 'oracle', 'lsap.c'), '-lm'])
    return ctypes.CDLL(so)


def _lsap(lib, c, maximize):
    n = c.shape[0]
    c = np.ascontiguousarray(c, dtype=np.float64)
    r = np.zeros(n, np.int32); cc = np.zeros(n, np.int32)
    rc = lib.lsap_oracle(c.ctypes.data_as(ctypes.c_void_p), n, int(maximize), r.ctypes.data_as(ctypes.c_void_p), cc.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0
    return r, cc


def test_lsap_oracle_bit_exact():
    """C restatement vs (a) fixtures solved by scipy on reference-built IoU matrices, (b) live scipy incl. ties."""
    lib = _lsap_lib()
    d = np.load(os.path.join(G, 'lsap.npz'))
    for i in range(int(d['count'])):
        r, c = _lsap(lib, d[f'cost{i}'], True)
        assert r.tolist() == d[f'row{i}'].tolist() and c.tolist() == d[f'col{i}'].tolist()
    from scipy.optimize import linear_sum_assignment
    rng = np.random.RandomState(1)
    for n in (1, 2, 4, 9, 13):
        for t in range(50):
            cost = rng.rand(n, n)
            if t % 2:
                cost = np.round(cost * 3) / 3
            for mx in (False, True):
                r, c = _lsap(lib, cost, mx)
                rs, cs = linear_sum_assignment(cost, maximize=mx)
                assert c.tolist() == cs.tolist()


def _metric_layouts():
    d = np.load(os.path.join(G, 'metrics.npz'))
    L1 = [(d[f'l1_b{i}'], d[f'l1_l{i}']) for i in range(int(d['n1']))]
    L2 = [(d[f'l2_b{i}'], d[f'l2_l{i}']) for i in range(int(d['n2']))]
    return d, L1, L2


def test_layout_metrics_oracle():
    """oracle/metrics_ref.py against the reference's own metric functions (golden from metrics/metric_layoutnet.py:66-150,204-242)."""
    from oracle import metrics_ref as R
    d, L1, L2 = _metric_layouts()
    assert np.abs(R.compute_iou(d['iou_in1'], d['iou_in2']) - d['iou']).max() <= 1e-7
    assert np.abs(R.compute_docsim_weight(d['iou_in1'], d['iou_in2']) - d['docsim_w']).max() <= 1e-7
    mi = np.asarray([R.compute_maximum_iou_for_layout(L1[i], L2[j]) for i, j in d['pairs']])
    md = np.asarray([R.compute_maximum_docsim_for_layout(L1[i], L2[j]) for i, j in d['pairs']])
    assert np.abs(mi - d['max_iou_pair']).max() <= 1e-7 and np.abs(md - d['max_docsim_pair']).max() <= 1e-7
    same = [(i, j) for i, j in d['pairs'] if (L1[i][1] == L2[j][1]).all()]
    pi = np.asarray([R.compute_iou_for_layout(L1[i], L2[j]) for i, j in same])
    assert np.abs(pi - d['iou_layout']).max() <= 1e-7
    assert abs(R.compute_maximum_iou(L1, L2) - float(d['max_iou_corpus'])) <= 1e-9


def test_layout_metric_formulas_product():
    """The product module's elementwise IoU / DocSim formulas (torch glue, device-agnostic) against the same golden."""
    from layoutdetr_amd.metrics import metric_layoutnet as M
    d, _, _ = _metric_layouts()
    assert np.abs(M.compute_iou(d['iou_in1'], d['iou_in2']) - d['iou']).max() <= 1e-7
    assert np.abs(M.compute_docsim_weight(d['iou_in1'], d['iou_in2']) - d['docsim_w']).max() <= 1e-7


@pytest.mark.parametrize('tag', ['', '_dh64'])
def test_bert_text_encoder_oracle(tag):
    """oracle/bert_ref.py against the outputs of the reference's BertEmbeddings + BertEncoder in text mode (training/med.py)."""
    from oracle import bert_ref
    d = np.load(os.path.join(G, f'bert_text{tag}.npz'))
    sd = {k[3:]: torch.from_numpy(d[k]) for k in d.files if k.startswith('sd/')}
    out = bert_ref.bert_text_forward(sd, int(d['num_heads']), torch.from_numpy(d['input_ids']), torch.from_numpy(d['attention_mask']))
    close(out, d['last_hidden_state'], 2e-6)


def test_bert_lm_decoder_oracle():
    """oracle/bert_ref.bert_lm_loss against the reference's text-mode LM decoder pieces: loss, logits and every gradient."""
    from oracle import bert_ref
    d = np.load(os.path.join(G, 'bert_lm.npz'))
    sd = {k[3:]: torch.from_numpy(d[k]).clone().requires_grad_(True) for k in d.files if k.startswith('sd/')}
    loss, logits = bert_ref.bert_lm_loss(sd, int(d['num_heads']), torch.from_numpy(d['input_ids']), torch.from_numpy(d['attention_mask']),
                                         torch.from_numpy(d['labels']))
    loss.backward()
    assert abs(loss.item() - float(d['loss'])) <= 1e-6
    close(logits.detach(), d['logits'], 2e-6)
    gmax = max(float(np.abs(d['grad/' + k]).max()) for k in sd)
    for k in sd:   # absolute, against the largest gradient: the key biases' true gradient is 0 (softmax shift invariance), i.e. noise
        assert (sd[k].grad - torch.from_numpy(d['grad/' + k])).abs().max().item() <= 2e-6 * gmax, k


def test_background_resample_oracle_bit_exact():
    """oracle/resample_ref.py against Pillow's own Lanczos resize (uint8, bit-exact) and the reference's two normalisation lines
    (fp32, bit-exact) on the committed fixtures: noise and page-like images, down- and up-scaling, ragged and equal sizes."""
    from oracle import resample_ref
    d = np.load(os.path.join(G, 'resample.npz'))
    for i in range(int(d['n'])):
        img = d[f'in{i}']
        s = d[f'u8_{i}'].shape[0]
        u8 = resample_ref.resize_antialias_u8(img, s, s)
        assert np.array_equal(u8, d[f'u8_{i}']), f'case {i}: uint8 resize differs'
        out = resample_ref.background_to_tensor(img, s)
        assert out.dtype == np.float32 and out.shape == (3, s, s)
        assert np.array_equal(out, d[f'out{i}']), f'case {i}: normalised tensor differs'


def test_box_ops_oracle_matches_reference_golden_bit_exact():
    """north_star row ns-1 (detr_util/box_ops.py): the numpy restatement against the reference's own functions — xyxy conversion, IoU,
    union, GIoU bit-exact in fp32; the Hungarian assignment on cost = -GIoU index-exact through the C restatement of scipy's solver."""
    from oracle import box_ops_ref
    d = np.load(os.path.join(G, 'box_ops.npz'))
    lib = _lsap_lib()
    lsap = lambda cost: _lsap(lib, cost, False)
    for i in range(int(d['count'])):
        p_xyxy = box_ops_ref.box_cxcywh_to_xyxy(d[f'pred{i}'])
        assert np.array_equal(p_xyxy, d[f'p_xyxy{i}'])
        assert np.array_equal(box_ops_ref.box_xyxy_to_cxcywh(p_xyxy), d[f'back{i}'])
        t_xyxy = box_ops_ref.box_cxcywh_to_xyxy(d[f'tgt{i}'])
        iou, union = box_ops_ref.box_iou(p_xyxy, t_xyxy)
        assert np.array_equal(iou, d[f'iou{i}'], equal_nan=True) and np.array_equal(union, d[f'union{i}'])
        assert np.array_equal(box_ops_ref.generalized_box_iou(p_xyxy, t_xyxy), d[f'giou{i}'], equal_nan=True)
        (r, c), _ = box_ops_ref.hungarian_match_giou(d[f'pred{i}'], d[f'tgt{i}'], lsap)
        assert r.tolist() == d[f'row{i}'].tolist() and c.tolist() == d[f'col{i}'].tolist(), i
    iou, union = box_ops_ref.box_iou(d['rect_a'], d['rect_b'])
    assert np.array_equal(iou, d['rect_iou']) and np.array_equal(union, d['rect_union'])
    assert np.array_equal(box_ops_ref.generalized_box_iou(d['rect_a'], d['rect_b']), d['rect_giou'])


# ---------------------------------------------------------------------------------------------------------------------
# Rows a13 / a14: the composition.  tests/golden/composition.npz = outputs of the reference's own Generator.forward,
# Discriminator.forward and StyleGAN2Loss.accumulate_gradients (oracle/gen_golden.py:gen_composition); weights and inputs
# are rebuilt from their names (oracle/seeded.py), the fixture holds expected values only.

COMP_SKIP = ('backbone.0.body.', 'text_encoder.', 'text_decoder.')


def comp_modules(bg):
    """Product-named state dicts for the oracle: the product modules are constructed on CPU for their parameter names and
    shapes only (no forward runs here), then every entry is overwritten by the seeded rule."""
    from layoutdetr_amd.training.networks_detr import Discriminator, Generator
    from oracle import seeded
    kw = dict(num_bbox_labels=8, img_channels=3, img_height=bg, img_width=bg, c_dim=0, background_size=bg, bert_f_dim=768, im_f_dim=512)
    Gm, Dm = Generator(z_dim=4, **kw), Discriminator(**kw)
    return Gm, Dm, seeded.seeded_state_dict(Gm, 1, COMP_SKIP), seeded.seeded_state_dict(Dm, 2, COMP_SKIP)


def comp_keys(sd):
    return sorted(f'{k}:{"x".join(map(str, v.shape))}' for k, v in sd.items() if not k.startswith(COMP_SKIP))


def digest_errors(g, d, phase, name):
    """(error of g, error of the reference's own fp32 run), both against the reference's fp64 run, relative to the largest entry
    of the fp64 gradient: the yardstick that separates fp32 summation noise from a real discrepancy."""
    from oracle import seeded
    st, sb = seeded.grad_digest(g)
    s32 = np.asarray(d[f'{phase}/gsub/{name}']).astype(np.float64); s64 = np.asarray(d[f'{phase}/gsub64/{name}'])
    st64 = np.asarray(d[f'{phase}/gstat64/{name}'])
    mx = float(st64[2]) + 1e-300
    e = max(float(np.abs(sb - s64).max()) / mx, abs(st[0] - float(st64[0])) / (float(st64[0]) + 1e-300))
    return e, float(np.abs(s32 - s64).max()) / mx


def test_composition_state_dict_names_match_reference():
    d = load('composition')
    Gm, Dm, Gsd, Dsd = comp_modules(int(d['bg']))
    assert comp_keys(Gsd) == sorted(d['G_keys'].tolist())
    assert comp_keys(Dsd) == sorted(d['D_keys'].tolist())


def test_composition_forward_tuples_vs_reference():
    from oracle import networks_ref, seeded
    d = load('composition')
    B, bg, seed = int(d['B']), int(d['bg']), int(d['seed'])
    inp = seeded.comp_inputs(B, bg, seed)
    _, _, Gsd, Dsd = comp_modules(bg)
    tf, tl, pm = d['text_feat'], d['text_len'], inp['padding_mask']
    with torch.no_grad():
        out = networks_ref.generator(Gsd, inp['z_g'], inp['bbox_class'], tf, tl, pm, inp['background'], reconst=True, feats=inp['feats_g'])
        for k, v in zip(('bbox_fake', 'loss_z', 'logit_cls', 'loss_lm', 'loss_text_len'), out):
            close(v, d['G/' + k], 1e-5)
        close(networks_ref.generator(Gsd, inp['z_g'], inp['bbox_class'], tf, tl, pm, inp['background'], feats=inp['feats_g']), d['G/bbox_fake_noreconst'], 1e-5)
        out = networks_ref.discriminator(Dsd, inp['bbox_real'], inp['bbox_class'], tf, tl, pm, inp['background'], reconst=True, bg_size=bg, feats=inp['feats_d'])
        for k, v in zip(('logit', 'logit_uncond', 'bbox_pred', 'logit_cls', 'loss_lm', 'loss_text_len', 'bg_rec', 'bbox_pred_uncond', 'logit_cls_uncond'), out):
            close(v, d['D/' + k], 2e-5)
        # ragged canvas (list of different-sized backgrounds): padding mask -> feature-map mask -> position encoding + attention masking
        bgl = [inp['background'][i, :, :h, :w] for i, (h, w) in enumerate(d['ragged_sizes'].tolist())]
        close(networks_ref.generator(Gsd, inp['z_g'], inp['bbox_class'], tf, tl, pm, bgl, feats=inp['feats_g']), d['G/bbox_fake_ragged'], 1e-5)
        lo = networks_ref.discriminator(Dsd, inp['bbox_real'], inp['bbox_class'], tf, tl, pm, bgl, feats=inp['feats_d'])
        close(lo[0], d['D/logit_ragged'], 1e-5); close(lo[1], d['D/logit_uncond_ragged'], 1e-5)
        assert not torch.equal(d['D/logit_ragged'], d['D/logit']), 'the ragged case must exercise the mask'


def test_config0_generator_forward_vs_reference():
    """BASELINE.json configs[0]: one sample, 128x128 background (16 memory tokens), 3 valid text boxes of 9 slots — the reference's own
    Generator.forward (oracle/gen_golden.py:gen_config0) against the oracle."""
    from oracle import networks_ref, seeded
    d = load('config0')
    B, bg, seed, nv = int(d['B']), int(d['bg']), int(d['seed']), int(d['nvalid'])
    assert (B, bg, nv) == (1, 128, 3)
    inp = seeded.comp_inputs(B, bg, seed)
    pm = torch.ones(B, 9, dtype=torch.bool); pm[:, :nv] = False
    _, _, Gsd, _ = comp_modules(bg)
    with torch.no_grad():
        close(networks_ref.generator(Gsd, inp['z_g'], inp['bbox_class'], d['text_feat'], d['text_len'], pm, inp['background'], feats=inp['feats_g']),
              d['G/bbox_fake_noreconst'], 1e-5)
        out = networks_ref.generator(Gsd, inp['z_g'], inp['bbox_class'], d['text_feat'], d['text_len'], pm, inp['background'], reconst=True, feats=inp['feats_g'])
        for k, v in zip(('bbox_fake', 'loss_z', 'logit_cls', 'loss_lm', 'loss_text_len'), out):
            close(v, d['G/' + k], 1e-5)
        assert out[2].shape[0] == nv


def test_composition_loss_phases_vs_reference():
    """StyleGAN2Loss.accumulate_gradients('Gmain' / 'Dmain'): every reported term and every parameter gradient."""
    from oracle import seeded, step_ref
    d = load('composition')
    B, bg, seed = int(d['B']), int(d['bg']), int(d['seed'])
    inp = seeded.comp_inputs(B, bg, seed)
    Gm, Dm, Gsd, Dsd = comp_modules(bg)
    fg, fd = inp['feats_g'].clone().requires_grad_(True), inp['feats_d'].clone().requires_grad_(True)
    bt = dict(bbox_real=inp['bbox_real'], bbox_class=inp['bbox_class'], text_feat=d['text_feat'], text_len=d['text_len'],
              padding_mask=inp['padding_mask'], background=inp['background'], feats_G=fg, feats_D=fd)
    out, gG, gD, _, _ = step_ref.training_iteration(Gsd, Dsd, bt, inp['z_g'], inp['z_d'], bg_size=bg, apply_adam=False,
                                                    G_param_names={n for n, _ in Gm.named_parameters()},
                                                    D_param_names={n for n, _ in Dm.named_parameters()})
    for phase, terms in (('Gmain', out['terms_G']), ('Dmain', out['terms_D'])):
        want = {k[len(phase) + 8:]: v for k, v in d.items() if k.startswith(phase + '/report/')}
        assert set(want) == set(terms), set(want) ^ set(terms)
        for k, v in want.items():
            close(terms[k], v, 2e-5)
    gG['backbone.0.body.feats'] = fg.grad; gD['backbone.0.body.feats'] = fd.grad
    worst = ref_worst = 0.0
    for phase, grads in (('Gmain', gG), ('Dmain', gD)):
        names = [k[len(phase) + 7:] for k in d if k.startswith(phase + '/gstat/')]
        body = [n for n in grads if n.startswith('backbone.0.body.') and n != 'backbone.0.body.feats']
        assert set(names) == set(grads) - set(body), set(names) ^ (set(grads) - set(body))
        for n in names:
            e, e_ref = digest_errors(grads[n], d, phase, n)
            # as close to the fp64 values as the reference's own fp32 run is (x3 for the luck of the sampled entries), floor 1e-5
            assert e <= max(3 * e_ref, 1e-5), f'{phase} {n}: {e:.3e} vs fp64 (reference fp32 run: {e_ref:.3e})'
            worst, ref_worst = max(worst, e), max(ref_worst, e_ref)
    print(f'composition gradients vs the fp64 reference run: oracle worst {worst:.2e}, reference fp32 run worst {ref_worst:.2e}')


def _reg_setup():
    from oracle import seeded
    d = load('reg')
    B, bg, seed = int(d['B']), int(d['bg']), int(d['seed'])
    inp = seeded.comp_inputs(B, bg, seed)
    Gm, Dm, Gsd, Dsd = comp_modules(bg)
    return d, inp, Gm, Dm, Gsd, Dsd


def test_regulariser_phases_vs_reference():
    """Path length ('Greg') and R1 of the reference's own StyleGAN2Loss.accumulate_gradients (tests/golden/reg.npz: second-order autograd
    through the reference's modules): every reported value, the running path-length mean, and every parameter gradient as close to
    the reference's fp64 run as its own fp32 run is."""
    from oracle import step_ref
    d, inp, Gm, Dm, Gsd, Dsd = _reg_setup()
    tf, tl = d['text_feat'], d['text_len']
    bg = int(d['bg'])
    fg, fd = inp['feats_g'].clone().requires_grad_(True), inp['feats_d'].clone().requires_grad_(True)
    bt = dict(bbox_real=inp['bbox_real'], bbox_class=inp['bbox_class'], text_feat=tf, text_len=tl, padding_mask=inp['padding_mask'],
              background=inp['background'], feats_G=fg, feats_D=fd)
    G = step_ref._params(Gsd, {n for n, _ in Gm.named_parameters()}); D = step_ref._params(Dsd, {n for n, _ in Dm.named_parameters()})
    # ---- path length
    t = {}
    loss, new_mean = step_ref.g_pl_loss(G, bt, inp['z_g'], d['pl_noise'], torch.zeros(()), float(d['pl_weight']), terms=t)
    (loss * 4).backward()
    close(t['Loss/pl_penalty'], d['Greg/report/Loss/pl_penalty'], 2e-5); close(t['Loss/G/reg'], d['Greg/report/Loss/G/reg'], 2e-5)
    close(new_mean, d['Greg/pl_mean'], 2e-6)
    gG = {k: v.grad for k, v in G.items() if v.requires_grad and v.grad is not None}
    gG['backbone.0.body.feats'] = fg.grad
    names = [k[len('Greg/gstat/'):] for k in d if k.startswith('Greg/gstat/')]
    assert set(names) == {n for n in gG if not (n.startswith('backbone.0.body.') and n != 'backbone.0.body.feats')}
    worst = ref_worst = 0.0
    for n in names:
        e, e_ref = digest_errors(gG[n], d, 'Greg', n)
        assert e <= max(3 * e_ref, 1e-5), f'Greg {n}: {e:.3e} vs fp64 (reference fp32 run: {e_ref:.3e})'
        worst, ref_worst = max(worst, e), max(ref_worst, e_ref)
    # ---- R1
    t = {}
    step_ref.d_r1_loss(D, bt, float(d['r1_gamma']), terms=t).backward()
    close(t['Loss/r1_penalty'], d['Dboth/report/Loss/r1_penalty'], 2e-5); close(t['Loss/D/reg'], d['Dboth/report/Loss/D/reg'], 2e-5)
    close(t['Loss/scores/real'], d['Dboth/report/Loss/scores/real'], 2e-5)
    gD = {k: v.grad for k, v in D.items() if v.requires_grad and v.grad is not None}
    gD['backbone.0.body.feats'] = fd.grad
    # the fixture holds grad('Dboth') - grad('Dmain') for EVERY parameter of D: R1's own gradient where R1 reaches (non-zero in fp64),
    # rounding residue of the subtraction elsewhere.  The oracle must reach exactly the parameters whose fp64 difference is non-zero.
    reach = {k[len('R1/gstat64/'):] for k in d if k.startswith('R1/gstat64/') and float(d[k][2]) > 0}
    got = {n for n in gD if not (n.startswith('backbone.0.body.') and n != 'backbone.0.body.feats')}
    assert got == reach, got ^ reach
    for n in sorted(got):
        e, e_ref = digest_errors(gD[n], d, 'R1', n)
        assert e <= max(3 * e_ref, 1e-5), f'R1 {n}: {e:.3e} vs fp64 (reference fp32 difference: {e_ref:.3e})'
        worst, ref_worst = max(worst, e), max(ref_worst, e_ref)
    print(f'regulariser gradients vs the fp64 reference run: oracle worst {worst:.2e}, reference fp32 run worst {ref_worst:.2e}')
