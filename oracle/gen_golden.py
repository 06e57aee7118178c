"""Generates tests/golden/*.npz by importing the REFERENCE (salesforce/LayoutDETR at /root/reference).

Run in the build container only:  python oracle/gen_golden.py
Nothing here is imported at test time; the fixtures are plain data (inputs, weights, expected
outputs and gradients) that pin the CPU oracle (oracle/*.py) to the reference's arithmetic.
Modules the reference cannot import here (torchvision, pytorch_fid, skimage are absent) are stubbed in
sys.modules with the minimum surface the import needs; none of the stubs takes part in a computation
that is captured.
"""
import os
import sys
import types

import numpy as np
import torch

REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')


def _stub_modules():
    tv = types.ModuleType('torchvision'); tv.__version__ = '0.13.1'; tv._is_tracing = lambda: False
    ops = types.ModuleType('torchvision.ops'); boxes = types.ModuleType('torchvision.ops.boxes')
    boxes.box_area = lambda b: (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    models = types.ModuleType('torchvision.models'); mu = types.ModuleType('torchvision.models._utils')
    mu.IntermediateLayerGetter = object
    tvu = types.ModuleType('torchvision.utils'); tvt = types.ModuleType('torchvision.transforms')
    tv.ops = ops; ops.boxes = boxes; tv.models = models; models._utils = mu; tv.utils = tvu; tv.transforms = tvt
    tv.__path__ = []
    for name, mod in [('torchvision', tv), ('torchvision.ops', ops), ('torchvision.ops.boxes', boxes),
                      ('torchvision.models', models), ('torchvision.models._utils', mu),
                      ('torchvision.utils', tvu), ('torchvision.transforms', tvt)]:
        sys.modules[name] = mod
    pf = types.ModuleType('pytorch_fid'); fs = types.ModuleType('pytorch_fid.fid_score')
    fs.calculate_frechet_distance = lambda *a, **k: 0.0
    pf.fid_score = fs
    sys.modules['pytorch_fid'] = pf; sys.modules['pytorch_fid.fid_score'] = fs
    sk = types.ModuleType('skimage'); skt = types.ModuleType('skimage.transform'); skt.resize = None
    sk.transform = skt
    sys.modules['skimage'] = sk; sys.modules['skimage.transform'] = skt


def npy(t):
    return t.detach().cpu().numpy()


def save(name, d):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **{k: (npy(v) if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in d.items()})
    print('wrote', path, f'{os.path.getsize(path) / 1024:.0f} KiB')


def gen_ops():
    from torch_utils.ops import bias_act, conv2d_resample, upfirdn2d
    from training.networks_stylegan2 import modulated_conv2d
    d = {}
    torch.manual_seed(100)
    x = torch.randn(2, 6, 5, 4) * 2; b = torch.randn(6)
    d['ba_x'] = x; d['ba_b'] = b
    for act in bias_act.activation_funcs.keys():
        xr = x.clone().requires_grad_(True); br = b.clone().requires_grad_(True)
        y = bias_act.bias_act(xr, br, act=act, impl='ref')
        g = torch.ones_like(y) * 0.5 + y.detach() * 0.1
        y.backward(g)
        d[f'ba_{act}_y'] = y; d[f'ba_{act}_dx'] = xr.grad; d[f'ba_{act}_db'] = br.grad
        d[f'ba_{act}_yc'] = bias_act.bias_act(x, b, act=act, alpha=0.1, gain=0.7, clamp=0.9, impl='ref')
    f = upfirdn2d.setup_filter([1, 3, 3, 1])
    d['f'] = f
    xu = torch.randn(2, 4, 7, 6); d['up_x'] = xu
    cases = [dict(up=1, down=1, padding=[1, 1, 1, 1], gain=4), dict(up=2, down=1, padding=[2, 1, 2, 1], gain=4),
             dict(up=1, down=2, padding=[1, 1, 1, 1], gain=1), dict(up=[2, 1], down=[1, 3], padding=[0, 2, -1, 3], gain=0.5),
             dict(up=3, down=2, padding=[-1, 4, 2, 0], gain=1.5, flip_filter=True)]
    fa = f + 0.01 * torch.arange(16.).reshape(4, 4)
    d['fa'] = fa
    for i, c in enumerate(cases):
        xr = xu.clone().requires_grad_(True)
        y = upfirdn2d.upfirdn2d(xr, fa, impl='ref', **c)
        y.backward(torch.ones_like(y) + 0.1 * y.detach())
        d[f'up{i}_y'] = y; d[f'up{i}_dx'] = xr.grad
    d['up2d_y'] = upfirdn2d.upsample2d(xu, f, impl='ref')
    xc = torch.randn(2, 8, 6, 6); w3 = torch.randn(12, 8, 3, 3) * 0.2; w1 = torch.randn(3, 8, 1, 1) * 0.3
    d['cr_x'] = xc; d['cr_w3'] = w3; d['cr_w1'] = w1
    d['cr_up2'] = conv2d_resample.conv2d_resample(xc, w3, f=f, up=2, padding=1, flip_weight=False)
    d['cr_up1'] = conv2d_resample.conv2d_resample(xc, w3, f=f, up=1, padding=1, flip_weight=True)
    d['cr_1x1'] = conv2d_resample.conv2d_resample(xc, w1, f=None, up=1, padding=0, flip_weight=True)
    d['cr_down2'] = conv2d_resample.conv2d_resample(xc, w3, f=f, down=2, padding=1)
    s = torch.randn(2, 8) + 1.0; d['mc_s'] = s
    for nm, kw in [('up2', dict(up=2, padding=1, resample_filter=f, flip_weight=False)), ('up1', dict(up=1, padding=1, flip_weight=True))]:
        xr = xc.clone().requires_grad_(True); wr = w3.clone().requires_grad_(True); sr = s.clone().requires_grad_(True)
        y = modulated_conv2d(x=xr, weight=wr, styles=sr, fused_modconv=False, **kw)
        y.backward(torch.ones_like(y) + 0.1 * y.detach())
        d[f'mc_{nm}_y'] = y; d[f'mc_{nm}_dx'] = xr.grad; d[f'mc_{nm}_dw'] = wr.grad; d[f'mc_{nm}_ds'] = sr.grad
    d['mc_rgb_y'] = modulated_conv2d(x=xc, weight=w1, styles=s, demodulate=False, fused_modconv=False)
    save('ops', d)


def gen_transformer():
    from training.detr_transformer import Transformer, TransformerWithToken
    from training.detr_position_encoding import PositionEmbeddingSine
    from detr_util.misc import NestedTensor
    from training.util import TransformerWithToken_layoutganpp
    B, d_model, nhead, h, w, Lq = 2, 64, 2, 3, 4, 5
    for name, cls in [('transformer', Transformer), ('transformer_token', TransformerWithToken)]:
        torch.manual_seed(200)
        m = cls(d_model=d_model, nhead=nhead, num_encoder_layers=2, num_decoder_layers=2, dim_feedforward=128, dropout=0.1).eval()
        src = torch.randn(B, d_model, h, w, requires_grad=True)
        mask = torch.zeros(B, h, w, dtype=torch.bool); mask[1, :, 3] = True
        pe = PositionEmbeddingSine(d_model // 2, normalize=True)
        pos = pe(NestedTensor(src, mask))
        tgt = torch.randn(Lq, B, d_model, requires_grad=True)
        kpm = torch.zeros(B, Lq, dtype=torch.bool); kpm[0, 3:] = True
        hs, mem = m(src, mask, pos, tgt, kpm)
        g_hs = torch.randn_like(hs); g_mem = torch.randn_like(mem) * 0.1
        (hs * g_hs).sum().add((mem * g_mem).sum()).backward()
        d = {'src': src, 'mask': mask, 'pos': pos, 'tgt': tgt, 'kpm': kpm, 'hs': hs, 'mem': mem, 'g_hs': g_hs, 'g_mem': g_mem,
             'd_src': src.grad, 'd_tgt': tgt.grad}
        for k, v in m.state_dict().items():
            d['sd/' + k] = v
        for k, p in m.named_parameters():
            if p.grad is not None:
                d['grad/' + k] = p.grad
        save(name, d)
    # nn.TransformerEncoder-based unconditional branch (training/util.py:13-43)
    torch.manual_seed(201)
    m = TransformerWithToken_layoutganpp(d_model=d_model, nhead=nhead, dim_feedforward=128, num_layers=2).eval()
    x = torch.randn(Lq, B, d_model, requires_grad=True)
    kpm = torch.zeros(B, Lq, dtype=torch.bool); kpm[1, 2:] = True
    y = m(x, src_key_padding_mask=kpm)
    g = torch.randn_like(y); (y * g).sum().backward()
    d = {'x': x, 'kpm': kpm, 'y': y, 'g': g, 'd_x': x.grad}
    for k, v in m.state_dict().items():
        d['sd/' + k] = v
    for k, p in m.named_parameters():
        d['grad/' + k] = p.grad
    save('transformer_layoutganpp', d)
    # position encoding at the real width (128 feats/axis) on a ragged mask
    pe = PositionEmbeddingSine(128, normalize=True)
    mask = torch.zeros(2, 4, 5, dtype=torch.bool); mask[0, 3:, :] = True; mask[1, :, 4:] = True
    save('pos_encoding', {'mask': mask, 'pos': pe(NestedTensor(torch.zeros(2, 1, 4, 5), mask))})


def gen_losses():
    from metrics.metric_layoutnet import compute_alignment, compute_overlap, generalized_iou_loss
    torch.manual_seed(300)
    B, N = 4, 9
    bbox = torch.cat([torch.rand(B, N, 2) * 0.6 + 0.2, torch.rand(B, N, 2) * 0.35 + 0.05], -1)
    bbox[0, 1] = bbox[0, 0]  # an exactly aligned / fully overlapping pair
    mask = torch.ones(B, N, dtype=torch.bool); mask[1, 5:] = False; mask[2, 1:] = False
    real = torch.cat([torch.rand(B, N, 2) * 0.6 + 0.2, torch.rand(B, N, 2) * 0.35 + 0.05], -1)
    d = {'bbox': bbox, 'mask': mask, 'real': real}
    for nm, fn in [('overlap', lambda b: compute_overlap(b, mask)), ('alignment', lambda b: compute_alignment(b, mask))]:
        br = bbox.clone().requires_grad_(True)
        v = fn(br); v.sum().backward()
        d[nm] = v; d['d_' + nm] = br.grad
    br = bbox.clone().requires_grad_(True)
    v = generalized_iou_loss(br[mask], real[mask]); v.backward()
    d['giou'] = v; d['d_giou'] = br.grad
    save('losses', d)


def gen_decoder():
    from training.networks_stylegan2 import Decoder
    torch.manual_seed(400)
    m = Decoder(z_dim=32, w_dim=32, img_resolution=16, img_channels=3, use_noise=False, channel_base=256, channel_max=32,
                num_fp16_res=0, conv_clamp=None, fused_modconv_default=False).train()
    z = torch.randn(2, 32, requires_grad=True)
    img = m(z)
    g = torch.randn_like(img); (img * g).sum().backward()
    d = {'z': z, 'img': img, 'g': g, 'd_z': z.grad}
    for k, v in m.state_dict().items():
        d['sd/' + k] = v
    for k, p in m.named_parameters():
        d['grad/' + k] = p.grad
    save('decoder', d)


def gen_frozen_bn():
    from training.detr_backbone import FrozenBatchNorm2d
    torch.manual_seed(500)
    bn = FrozenBatchNorm2d(6)
    bn.weight.copy_(torch.rand(6) + 0.5); bn.bias.copy_(torch.randn(6)); bn.running_mean.copy_(torch.randn(6)); bn.running_var.copy_(torch.rand(6) + 0.2)
    x = torch.randn(2, 6, 3, 3)
    save('frozen_bn', {'x': x, 'y': bn(x), 'weight': bn.weight, 'bias': bn.bias, 'running_mean': bn.running_mean, 'running_var': bn.running_var})


def gen_lsap():
    """Cost matrices built as metrics/metric_layoutnet.py:100-113 builds them (pairwise IoU of same-label boxes),
    solved by scipy.optimize.linear_sum_assignment(maximize=True) — the only Hungarian arithmetic in the reference."""
    import scipy
    from scipy.optimize import linear_sum_assignment
    from metrics.metric_layoutnet import compute_iou
    rng = np.random.RandomState(7)
    d = {'scipy_version': scipy.__version__}
    idx = 0
    for n in [1, 2, 3, 4, 5, 7, 9]:
        for rep in range(6):
            bi = np.concatenate([rng.rand(n, 2) * 0.6 + 0.2, rng.rand(n, 2) * 0.3 + 0.05], 1)
            bj = np.concatenate([rng.rand(n, 2) * 0.6 + 0.2, rng.rand(n, 2) * 0.3 + 0.05], 1)
            if rep == 1:
                bj = bi.copy()            # identical layouts: diagonal ones, ties elsewhere
            if rep == 2:
                bj = bi[::-1].copy()
            if rep == 3:
                bi[:] = bi[0]; bj[:] = bj[0]   # constant matrix -> identity assignment
            ii, jj = np.meshgrid(range(n), range(n)); ii, jj = ii.flatten(), jj.flatten()
            iou = compute_iou(bi[ii], bj[jj]).reshape(n, n)
            if rep == 4:
                iou = np.round(iou * 4) / 4   # quantised -> many ties
            if rep == 5:
                iou = np.zeros((n, n))
            r, c = linear_sum_assignment(iou, maximize=True)
            d[f'cost{idx}'] = iou; d[f'row{idx}'] = r; d[f'col{idx}'] = c
            idx += 1
    d['count'] = idx
    save('lsap', d)


def gen_metrics():
    """Layout metrics of the evaluation path: pairwise IoU / DocSim weights, per-pair maximum scores with the
    per-label Hungarian matching, and the corpus-level compute_maximum_iou (metrics/metric_layoutnet.py:66-150, 204-242)."""
    import numpy as np
    from metrics import metric_layoutnet as M
    rng = np.random.RandomState(77)

    def layout(labels):
        n = len(labels)
        b = np.concatenate([rng.rand(n, 2) * 0.6 + 0.2, rng.rand(n, 2) * 0.35 + 0.05], -1).astype(np.float32)
        return b, np.asarray(labels, dtype=np.int64)

    conds = [[0, 0, 1, 2, 2, 2, 3], [1, 1, 1, 1], [0, 1, 2, 3, 4, 5, 6, 7, 0], [2], [3, 3, 0]]
    L1, L2 = [], []
    for c in conds:
        for _ in range(3):
            L1.append(layout(list(rng.permutation(c))))
        for _ in range(4):
            L2.append(layout(list(rng.permutation(c))))
    L2.append(layout([4, 4]))                     # a condition with no counterpart
    L1[1] = (L1[0][0].copy(), L1[0][1].copy())    # exact duplicate layout -> tied assignments
    d = {}
    for i, (b, l) in enumerate(L1):
        d[f'l1_b{i}'] = b; d[f'l1_l{i}'] = l
    for i, (b, l) in enumerate(L2):
        d[f'l2_b{i}'] = b; d[f'l2_l{i}'] = l
    d['n1'] = np.asarray(len(L1)); d['n2'] = np.asarray(len(L2))
    b1 = np.concatenate([b for b, _ in L1[:6]]); b2 = np.concatenate([b for b, _ in L2[:8]])[:len(b1)]
    d['iou_in1'] = b1; d['iou_in2'] = b2
    d['iou'] = M.compute_iou(b1, b2); d['docsim_w'] = M.compute_docsim_weight(b1, b2)
    # per-pair maximum scores for every same-condition pair, in (i, j) order of the lists above
    pairs, miou, mdoc, piou, pdoc = [], [], [], [], []
    for i, (bi, li) in enumerate(L1):
        for j, (bj, lj) in enumerate(L2):
            if sorted(li.tolist()) == sorted(lj.tolist()):
                pairs.append((i, j))
                miou.append(M.compute_maximum_iou_for_layout((bi, li), (bj, lj)))
                mdoc.append(M.compute_maximum_docsim_for_layout((bi, li), (bj, lj)))
                if (li == lj).all():
                    piou.append(M.compute_iou_for_layout((bi, li), (bj, lj))); pdoc.append(M.compute_docsim_for_layout((bi, li), (bj, lj)))
    d['pairs'] = np.asarray(pairs); d['max_iou_pair'] = np.asarray(miou); d['max_docsim_pair'] = np.asarray(mdoc)
    d['iou_layout'] = np.asarray(piou); d['docsim_layout'] = np.asarray(pdoc)
    d['max_iou_corpus'] = np.asarray(M.compute_maximum_iou(L1, L2, n_jobs=1))
    save('metrics', d)


def gen_bert():
    """Text-mode BERT encoder: outputs of the reference's own BertEmbeddings + BertEncoder (training/med.py).
    BertModel itself cannot be constructed under the installed transformers (5.x changed PreTrainedModel's init protocol), and
    med.py imports three helpers that moved since 4.19.2; they are aliased to their current homes (real functions of the same
    dependency) or, for the head-pruning helper that the forward never calls, to a stub that raises.  The mask preparation
    of BertModel.forward (:709-748) is the one-liner below."""
    hidden = {k: sys.modules.pop(k) for k in list(sys.modules) if k.split('.')[0] == 'torchvision'}   # transformers probes the real package
    import transformers.modeling_utils as mu
    import transformers.pytorch_utils as pu
    mu.apply_chunking_to_forward = pu.apply_chunking_to_forward
    mu.prune_linear_layer = pu.prune_linear_layer

    def _unused(*a, **k):
        raise NotImplementedError
    mu.find_pruneable_heads_and_indices = _unused
    from training.med import BertConfig, BertEmbeddings, BertEncoder
    sys.modules.update(hidden)
    for tag, (hid, heads, layers, inter, T, B) in {'': (64, 2, 2, 128, 12, 3), '_dh64': (128, 2, 2, 256, 21, 4)}.items():
        torch.manual_seed(410 + hid)
        cfg = BertConfig(vocab_size=60, hidden_size=hid, num_hidden_layers=layers, num_attention_heads=heads, intermediate_size=inter,
                         max_position_embeddings=40, add_cross_attention=True, encoder_width=hid)
        emb, enc = BertEmbeddings(cfg).eval(), BertEncoder(cfg).eval()
        for nm, prm in list(emb.named_parameters()) + list(enc.named_parameters()):
            prm.data.normal_(0, 0.08)
            if 'LayerNorm.weight' in nm:
                prm.data.add_(1.0)
        ids = torch.randint(1, 60, (B, T)); am = torch.ones(B, T, dtype=torch.long)
        am[1, T // 2:] = 0; am[B - 1, 1:] = 0; ids[am == 0] = 0
        ext = (1.0 - am[:, None, None, :].float()) * -10000.0
        with torch.no_grad():
            out = enc(emb(input_ids=ids), attention_mask=ext, return_dict=True, mode='text').last_hidden_state
        d = {'input_ids': ids, 'attention_mask': am, 'last_hidden_state': out, 'num_heads': np.asarray(heads)}
        for k, v in emb.state_dict().items():
            d['sd/embeddings.' + k] = v
        for k, v in enc.state_dict().items():
            if 'crossattention' not in k:      # never read in text mode; left out of the fixture
                d['sd/encoder.' + k] = v
        save('bert_text' + tag, d)


def gen_bert_lm():
    """Text-mode LM decoder (second half): the reference's BertEmbeddings + BertEncoder + BertOnlyMLMHead driven as
    BertLMHeadModel.forward drives them in mode='text' (causal x padding mask from get_extended_attention_mask, med.py:704-739;
    decoder weight tied to the word embeddings; shifted cross entropy with label_smoothing=0.1, :911-916), loss + the gradient of
    every parameter.  Same transformers-5.x workaround as gen_bert (the PreTrainedModel wrapper cannot be constructed)."""
    hidden = {k: sys.modules.pop(k) for k in list(sys.modules) if k.split('.')[0] == 'torchvision'}
    import transformers.modeling_utils as mu
    import transformers.pytorch_utils as pu
    mu.apply_chunking_to_forward = pu.apply_chunking_to_forward
    mu.prune_linear_layer = pu.prune_linear_layer

    def _unused(*a, **k):
        raise NotImplementedError
    mu.find_pruneable_heads_and_indices = _unused
    from training.med import BertConfig, BertEmbeddings, BertEncoder, BertOnlyMLMHead
    sys.modules.update(hidden)
    torch.manual_seed(520)
    hid, heads, layers, inter, T, B, V = 64, 2, 2, 128, 12, 4, 70
    cfg = BertConfig(vocab_size=V, hidden_size=hid, num_hidden_layers=layers, num_attention_heads=heads, intermediate_size=inter,
                     max_position_embeddings=40, add_cross_attention=True, encoder_width=hid, is_decoder=True)
    emb, enc, head = BertEmbeddings(cfg).eval(), BertEncoder(cfg).eval(), BertOnlyMLMHead(cfg).eval()
    head.predictions.decoder.weight = emb.word_embeddings.weight     # HF tie_weights
    params = {}
    for prefix, mod in (('bert.embeddings.', emb), ('bert.encoder.', enc), ('cls.', head)):
        for nm, prm in mod.named_parameters():
            if 'crossattention' in nm or (prefix == 'cls.' and nm == 'predictions.decoder.weight'):
                continue
            prm.data.normal_(0, 0.08)
            if 'LayerNorm.weight' in nm:
                prm.data.add_(1.0)
            params[prefix + nm] = prm
    ids = torch.randint(1, V, (B, T)); am = torch.ones(B, T, dtype=torch.long)
    am[1, 7:] = 0; am[3, 2:] = 0; ids[am == 0] = 0
    ids[:, 0] = V - 2                                      # "[DEC]" bos id, as networks_detr.py:172 writes it
    labels = ids.masked_fill(ids == 0, -100)
    labels[2] = -100                                       # a padded layout slot: contributes nothing
    causal = torch.tril(torch.ones(T, T))[None, None]
    ext = (1.0 - causal * am[:, None, None, :].float()) * -10000.0
    hs = enc(emb(input_ids=ids), attention_mask=ext, return_dict=True, mode='text').last_hidden_state
    scores = head(hs)[:, :-1, :].contiguous()
    loss = torch.nn.CrossEntropyLoss(reduction='mean', label_smoothing=0.1)(scores.view(-1, V), labels[:, 1:].contiguous().view(-1))
    loss.backward()
    d = {'input_ids': ids, 'attention_mask': am, 'labels': labels, 'loss': loss.detach(), 'logits': scores.detach(), 'num_heads': np.asarray(heads)}
    for k, prm in params.items():
        d['sd/' + k] = prm.detach(); d['grad/' + k] = prm.grad
    save('bert_lm', d)


def gen_resample():
    """Background preprocessing of a dataset item (dataset_layoutganpp.py:330-338): PIL Lanczos resize of a uint8 RGB page image,
    then (x / 255 - mean) / std in fp32, CHW.  The resize arithmetic is Pillow's (pinned 9.3.0, environment.yaml:46; 12.x here); the
    two numpy lines are evaluated exactly as the reference writes them."""
    import PIL
    from PIL import Image
    rng = np.random.default_rng(77)
    rgb_mean = np.reshape(np.array([0.485, 0.456, 0.406]).astype(np.float32), (1, 1, 3))
    rgb_std = np.reshape(np.array([0.229, 0.224, 0.225]).astype(np.float32), (1, 1, 3))
    d = {'pillow_version': np.array([int(v) for v in PIL.__version__.split('.')[:2]])}
    cases = [(64, 64, 16), (70, 100, 32), (96, 33, 24), (24, 20, 48), (48, 48, 48), (129, 65, 31)]   # (H, W, background_size)
    for i, (h, w, s) in enumerate(cases):
        if i % 2 == 0:
            img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)          # noise: exercises the clip8 overshoot of the negative lobes
        else:                                                               # smooth gradients + hard edges, like a rendered page
            yy, xx = np.mgrid[0:h, 0:w]
            img = np.stack([(yy * 255 // max(h - 1, 1)), (xx * 255 // max(w - 1, 1)), ((xx // 7 + yy // 5) % 2) * 255], -1).astype(np.uint8)
        lanczos = getattr(Image, 'ANTIALIAS', Image.LANCZOS)                # ANTIALIAS is the old name of LANCZOS
        background = np.array(Image.fromarray(img).resize((s, s), lanczos))
        d[f'in{i}'] = img
        d[f'u8_{i}'] = background
        background = (background.astype(np.float32) / 255.0 - rgb_mean) / rgb_std
        d[f'out{i}'] = background.transpose(2, 0, 1)
    d['n'] = np.array(len(cases))
    np.savez_compressed(os.path.join(OUT, 'resample.npz'), **d)
    print('wrote resample.npz')


def gen_dp_step():
    from torch_utils import misc
    torch.manual_seed(600)
    grads = [torch.randn(7), torch.randn(3, 4), torch.randn(2, 2, 2)]
    grads[0][1] = float('nan'); grads[1][0, 0] = float('inf'); grads[2][1, 1, 1] = -float('inf')
    d = {}
    for W in (1, 2, 8):
        flat = torch.cat([g.flatten() for g in grads]) * W   # what an all_reduce SUM of W identical ranks yields
        if W > 1:
            flat /= W
        misc.nan_to_num(flat, nan=0, posinf=1e5, neginf=-1e5, out=flat)
        d[f'out_w{W}'] = flat
    for i, g in enumerate(grads):
        d[f'g{i}'] = g
    save('dp_step', d)

def _import_ref_networks():
    """`training.networks_detr` of the reference, imported with the absent third-party packages stubbed: timm / fairscale are only
    touched by training/vit.py + blip.py at import time (class decorators, base-class helpers), never by Generator/Discriminator.forward."""
    hidden = {k: sys.modules.pop(k) for k in list(sys.modules) if k.split('.')[0] == 'torchvision'}
    import transformers.modeling_utils as mu
    import transformers.pytorch_utils as pu
    from transformers import BertTokenizer  # noqa: F401  (blip.py imports it)
    mu.apply_chunking_to_forward = pu.apply_chunking_to_forward
    mu.prune_linear_layer = pu.prune_linear_layer
    if not hasattr(mu, 'find_pruneable_heads_and_indices'):
        def _unused(*a, **k):
            raise NotImplementedError
        mu.find_pruneable_heads_and_indices = _unused
    sys.modules.update(hidden)

    def mod(name, **attrs):
        m = types.ModuleType(name); m.__path__ = []
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m
    ident = lambda *a, **k: (a[0] if a else None)
    mod('timm'); mod('timm.models')
    mod('timm.models.vision_transformer', _cfg=lambda **k: {}, PatchEmbed=torch.nn.Module)
    mod('timm.models.registry', register_model=lambda f: f)
    mod('timm.models.layers', trunc_normal_=ident, DropPath=torch.nn.Identity)
    mod('timm.models.helpers', named_apply=ident, adapt_input_conv=ident)
    mod('timm.models.hub', download_cached_file=ident)
    mod('fairscale'); mod('fairscale.nn'); mod('fairscale.nn.checkpoint')
    mod('fairscale.nn.checkpoint.checkpoint_activations', checkpoint_wrapper=ident)
    import training.networks_detr as nd
    return nd


class _Tok(object):
    """Stand-in for the BERT tokenizer (needs the bert-base-uncased vocabulary file: no network here).  Deterministic character
    hashing; only the container protocol Generator/Discriminator.forward uses (`input_ids`, `attention_mask`, `.to`)."""
    bos_token_id, pad_token_id, T = 30522, 0, 12

    def __call__(self, texts, padding=None, truncation=None, max_length=None, return_tensors=None):
        ids = torch.zeros(len(texts), self.T, dtype=torch.long); am = torch.zeros_like(ids)
        for i, t in enumerate(texts):
            toks = [101] + [1000 + (ord(c) * 7) % 20000 for c in t][:self.T - 2] + [102]
            ids[i, :len(toks)] = torch.tensor(toks); am[i, :len(toks)] = 1
        out = types.SimpleNamespace(input_ids=ids, attention_mask=am)
        out.to = lambda dev: out
        return out


class _TextEnc(torch.nn.Module):
    """Stand-in for the frozen BERT text encoder (pinned separately: bert_text*.npz): a fixed table lookup, so that the CLS
    feature handed to fc_in is a known function of the token ids."""

    def __init__(self):
        super().__init__()
        from oracle import seeded
        self.register_buffer('table', seeded.uniform('text_encoder.table', (97, 768), 0, -1.0, 1.0))

    def forward(self, input_ids, attention_mask=None, return_dict=True, mode='text'):
        h = self.table[(input_ids * 31 + torch.arange(input_ids.shape[1])[None]) % 97] * attention_mask[..., None].float()
        return types.SimpleNamespace(last_hidden_state=h.cumsum(1).flip(1))   # CLS position sees the whole text


class _TextDec(torch.nn.Module):
    """Stand-in for the LM text decoder (pinned separately: bert_lm.npz).  Returns a zero loss, as the product does in
    text_mode='features'; the arguments the reference builds for it are recorded for the wiring fixture."""

    def __init__(self):
        super().__init__()
        self.calls = []

    def forward(self, input_ids, attention_mask=None, encoder_hidden_states=None, encoder_attention_mask=None, labels=None,
                return_dict=True, mode='text'):
        self.calls.append(dict(input_ids=input_ids.clone(), attention_mask=attention_mask.clone(), labels=labels.clone()))
        return types.SimpleNamespace(loss=encoder_hidden_states.new_zeros(()))


class _Body(torch.nn.Module):
    """Stand-in for torchvision's ResNet-50 body behind IntermediateLayerGetter (absent here): a learnable feature map per
    canvas shape, so the gradient that reaches the trunk output is captured like any parameter gradient."""

    def __init__(self, feats):
        super().__init__()
        self.feats = torch.nn.Parameter(feats)

    def forward(self, x):
        assert x.shape[0] <= self.feats.shape[0] and x.shape[2] // 32 == self.feats.shape[2] and x.shape[3] // 32 == self.feats.shape[3]
        return {'0': self.feats[:x.shape[0]]}      # (the path-length phase runs the first half of the batch, loss.py:121-128)


def _ref_G_D(nd, bg, feats_g, feats_d):
    """Generator / Discriminator instances of the REFERENCE classes whose __init__ (network access, torchvision) is bypassed: the
    attributes are the reference's own sub-module classes with the constructor arguments of networks_detr.py:66-131 / 191-277."""
    from torch import nn
    from training.detr_backbone import BackboneBase, Joiner
    from training.detr_position_encoding import PositionEmbeddingSine
    from training.detr_transformer import Transformer, TransformerWithToken
    from training.networks_stylegan2 import Decoder
    from training.util import TransformerWithToken_layoutganpp
    hd, bf, L, T = 256, 768, 8, 256

    def backbone(feats):
        bb = BackboneBase.__new__(BackboneBase); nn.Module.__init__(bb)
        bb.body = _Body(feats); bb.num_channels = 2048
        j = Joiner(bb, PositionEmbeddingSine(num_pos_feats=128, normalize=True)); j.num_channels = 2048
        return j
    tr = dict(d_model=hd, dropout=0.1, nhead=8, dim_feedforward=2048, num_encoder_layers=6, num_decoder_layers=6, normalize_before=False,
              return_intermediate_dec=False)
    G = nd.Generator.__new__(nd.Generator); nn.Module.__init__(G)
    G.z_dim, G.num_bbox_labels, G.c_dim, G.max_text_length = 4, L, 0, T
    G.backbone = backbone(feats_g); G.input_proj = nn.Conv2d(2048, hd, kernel_size=1)
    G.fc_z = nn.Linear(4 * 9, bf); G.emb_label = nn.Embedding(L, bf)
    G.tokenizer = _Tok(); G.text_encoder = _TextEnc(); G.enc_text_len = nn.Embedding(T, bf)
    G.fc_in = nd.MLP(input_dim=4 * bf, hidden_dim=bf, output_dim=hd, num_layers=3)
    G.transformer = Transformer(**tr)
    G.bbox_embed = nd.MLP(input_dim=hd, hidden_dim=hd, output_dim=4, num_layers=3)
    G.fc_z_rec = nn.Linear(hd, 4 * 9); G.fc_out_cls = nn.Linear(hd, L); G.text_decoder = _TextDec(); G.fc_text_len_rec = nn.Linear(hd, T)

    D = nd.Discriminator.__new__(nd.Discriminator); nn.Module.__init__(D)
    D.num_bbox_labels, D.c_dim, D.max_text_length = L, 0, T
    D.backbone = backbone(feats_d); D.input_proj = nn.Conv2d(2048, hd, kernel_size=1)
    D.fc_bbox = nn.Linear(4, bf); D.emb_label = nn.Embedding(L, bf)
    D.tokenizer = _Tok(); D.text_encoder = _TextEnc(); D.enc_text_len = nn.Embedding(T, bf)
    D.enc_fc_in = nd.MLP(input_dim=4 * bf, hidden_dim=bf, output_dim=hd, num_layers=3)
    D.enc_transformer = TransformerWithToken(**tr)
    D.fc_out_disc = nn.Linear(hd, 1)
    D.pos_token = nn.Parameter(torch.rand(50, 1, hd)); D.dec_fc_in = nn.Linear(2 * hd, hd)
    D.dec_transformer = nn.TransformerEncoder(nn.TransformerEncoderLayer(d_model=hd, nhead=8, dim_feedforward=2048), num_layers=6)
    D.bbox_embed = nn.Linear(hd, 4); D.fc_out_cls = nn.Linear(hd, L); D.text_decoder = _TextDec(); D.fc_text_len_rec = nn.Linear(hd, T)
    D.bg_decoder = Decoder(z_dim=hd, w_dim=512, channel_max=512, channel_base=8192, img_channels=3, img_resolution=bg, use_noise=False,
                           num_fp16_res=0, conv_clamp=None, fused_modconv_default=False)
    D.fc_bbox_uncond = nn.Linear(4, bf); D.emb_label_uncond = nn.Embedding(L, bf)
    D.enc_fc_in_uncond = nd.MLP(input_dim=2 * bf, hidden_dim=bf, output_dim=hd, num_layers=3)
    D.enc_transformer_uncond = TransformerWithToken_layoutganpp(d_model=hd, dim_feedforward=2048, nhead=8, num_layers=6)
    D.fc_out_disc_uncond = nn.Linear(hd, 1)
    D.pos_token_uncond = nn.Parameter(torch.rand(50, 1, hd)); D.dec_fc_in_uncond = nn.Linear(2 * hd, hd)
    D.dec_transformer_uncond = nn.TransformerEncoder(nn.TransformerEncoderLayer(d_model=hd, nhead=8, dim_feedforward=2048), num_layers=6)
    D.bbox_embed_uncond = nn.Linear(hd, 4); D.fc_out_cls_uncond = nn.Linear(hd, L)
    return G.eval(), D.eval()      # eval: dropout off (numeric parity runs with dropout off)


def gen_composition():
    """Rows a13 / a14: the reference's OWN Generator.forward / Discriminator.forward (networks_detr.py:133-187, 279-361) and
    StyleGAN2Loss.accumulate_gradients (loss.py:75-218) at the real layer sizes (hidden 256, 6+6 layers, 8 heads, 768-wide text
    features, StyleGAN2 Decoder at the background size), driven on CPU.  Only the three pieces that cannot exist in this container
    are stand-ins (classes above): the torchvision ResNet-50 body (a learnable feature map: the gradient reaching the trunk is
    captured), the BERT tokenizer / text encoder (table lookup) and the LM decoder (zero loss; its call arguments are recorded).
    Weights and inputs are pure functions of their names (oracle/seeded.py), so the fixture holds expected values only."""
    from oracle import seeded
    nd = _import_ref_networks()
    from torch_utils import training_stats
    from training.loss import StyleGAN2Loss
    B, bg, seed = 3, 64, 11
    inp = seeded.comp_inputs(B, bg, seed)
    G, D = _ref_G_D(nd, bg, inp['feats_g'], inp['feats_d'])
    skip = ('backbone.0.body.', 'text_encoder.', 'text_decoder.')
    G.load_state_dict(seeded.seeded_state_dict(G, 1, skip)); D.load_state_dict(seeded.seeded_state_dict(D, 2, skip))
    d = {'B': np.asarray(B), 'bg': np.asarray(bg), 'seed': np.asarray(seed)}
    d['G_keys'] = np.asarray([f'{k}:{"x".join(map(str, v.shape))}' for k, v in G.state_dict().items() if not k.startswith(skip)])
    d['D_keys'] = np.asarray([f'{k}:{"x".join(map(str, v.shape))}' for k, v in D.state_dict().items() if not k.startswith(skip)])
    patch = torch.zeros(B, 9, 1, 1, 1)
    c = torch.zeros(B, 0)
    tok = G.tokenizer(sum(inp['texts'], []))
    d['input_ids'] = tok.input_ids; d['attention_mask'] = tok.attention_mask
    d['text_feat'] = G.text_encoder(tok.input_ids, tok.attention_mask).last_hidden_state[:, 0, :].view(B, 9, -1)
    d['text_len'] = torch.tensor([len(t) for t in sum(inp['texts'], [])]).view(B, 9)
    # ---- forward tuples (a13)
    with torch.no_grad():
        out = G(inp['z_g'], inp['bbox_class'], inp['bbox_real'], inp['texts'], patch, inp['padding_mask'], inp['background'], c, True)
        for k, v in zip(('bbox_fake', 'loss_z', 'logit_cls', 'loss_lm', 'loss_text_len'), out):
            d['G/' + k] = v
        d['G/bbox_fake_noreconst'] = G(inp['z_g'], inp['bbox_class'], inp['bbox_real'], inp['texts'], patch, inp['padding_mask'], inp['background'], c)
        out = D(inp['bbox_real'], inp['bbox_class'], inp['texts'], patch, inp['padding_mask'], inp['background'], c, True)
        for k, v in zip(('logit', 'logit_uncond', 'bbox_pred', 'logit_cls', 'loss_lm', 'loss_text_len', 'bg_rec', 'bbox_pred_uncond', 'logit_cls_uncond'), out):
            d['D/' + k] = v
        lo = D(inp['bbox_real'], inp['bbox_class'], inp['texts'], patch, inp['padding_mask'], inp['background'], c)
        d['D/logit_noreconst'] = lo[0]; d['D/logit_uncond_noreconst'] = lo[1]
        # LM-decoder wiring (networks_detr.py:169-181): what the reference hands to text_decoder
        call = G.text_decoder.calls[0]
        d['G/lm_input_ids'] = call['input_ids']; d['G/lm_attention_mask'] = call['attention_mask']; d['G/lm_labels'] = call['labels']
        # ragged canvas: a list of different-sized backgrounds (nested_tensor_from_tensor_list + mask interpolation, detr_backbone.py:82-95)
        sizes = [(64, 64), (32, 64), (64, 32)]
        bgl = [inp['background'][i, :, :h, :w].clone() for i, (h, w) in enumerate(sizes)]
        d['ragged_sizes'] = np.asarray(sizes)
        d['G/bbox_fake_ragged'] = G(inp['z_g'], inp['bbox_class'], inp['bbox_real'], inp['texts'], patch, inp['padding_mask'], bgl, c)
        lo = D(inp['bbox_real'], inp['bbox_class'], inp['texts'], patch, inp['padding_mask'], bgl, c)
        d['D/logit_ragged'] = lo[0]; d['D/logit_uncond_ragged'] = lo[1]
    # ---- loss phases (a14) exactly as training_loop.py:281-313 drives them: zero_grad, requires_grad_(True) on the phase's module only.
    # Run twice: fp32 (the reference's arithmetic; tag '') and fp64 (tag '64': the same reference code in double precision, the
    # yardstick that tells fp32 rounding noise from a real discrepancy: a path is right when it is as close to the fp64 values as
    # the reference's own fp32 run is).
    reports = {}
    real_report = training_stats.report
    training_stats.report = lambda name, value: reports.setdefault(name, []).append(torch.as_tensor(value).detach().clone()) or value
    import training.loss as loss_mod
    loss_mod.training_stats.report = training_stats.report
    for tag, dt in (('', torch.float32), ('64', torch.float64)):
        G.to(dt); D.to(dt)
        D.bg_decoder.float()      # the StyleGAN2 blocks cast their activations to fp32 themselves (networks_stylegan2.py:402-433): this sub-network stays fp32 in the '64' run
        cast = lambda t: t.to(dt) if t.dtype.is_floating_point else t
        loss = StyleGAN2Loss(torch.device('cpu'), G, D)        # default weights = train.py:263-275
        G.requires_grad_(False); D.requires_grad_(False)
        for phase, mod_ in (('Gmain', G), ('Dmain', D)):
            reports.clear()
            mod_.requires_grad_(True); mod_.text_encoder.requires_grad_(False)
            for p in mod_.parameters():
                p.grad = None
            loss.accumulate_gradients(phase=phase, bbox_real=cast(inp['bbox_real']), bbox_class=inp['bbox_class'], bbox_text=inp['texts'], bbox_patch=cast(patch),
                                      padding_mask=inp['padding_mask'], background=inp['background'], real_c=cast(c),   # background: only mse_loss(bg_rec, background) reads it (the body is a stand-in), and bg_rec is fp32
                                     
                                      gen_z=cast(inp['z_g'] if phase == 'Gmain' else inp['z_d']), gen_c=cast(c), gain=1, cur_nimg=0)
            mod_.requires_grad_(False)
            for name, vals in reports.items():
                for i, v in enumerate(vals):
                    d[f'{phase}/report{tag}/{name}' + (f'#{i}' if len(vals) > 1 else '')] = v
            for name, p in mod_.named_parameters():
                if p.grad is None:
                    continue
                st, sub = seeded.grad_digest(p.grad)
                d[f'{phase}/gstat{tag}/{name}'] = st; d[f'{phase}/gsub{tag}/{name}'] = sub if tag else sub.astype(np.float32)
    training_stats.report = real_report
    save('composition', d)


def gen_reg():
    """The regulariser terms of the reference's OWN StyleGAN2Loss.accumulate_gradients: path length ('Greg', loss.py:119-142, pl_weight = 2,
    pl_batch_shrink = 2) and R1 (loss.py:162-166 + 207-217, r1_gamma = 10) -- second-order autograd through the reference's Generator /
    Discriminator at the real layer sizes (same stand-ins and seeded weights / inputs as gen_composition), in fp32 (tag '') and fp64 ('64').
    R1 cannot be captured from phase 'Dreg': the reference raises UnboundLocalError there (loss.py:218 sums `loss_Dreal_text_len_rec`, which
    the zero-initialisation block :168-176 omits and only the `phase in ['Dmain', 'Dboth']` branch assigns) -- i.e. `--gamma > 0` with lazy
    regularisation (D_reg_interval = 16, the training_loop default train.py leaves in place) does not run in the reference.  It is captured
    from phase 'Dboth' (main terms + R1 on one forward pass, dropout off): reported values directly, parameter gradients as
    grad('Dboth') - grad('Dmain') -- the R1 term's own gradient.
    Stored: every reported value, the path-length noise draw (torch.randn_like at :131, captured), the running mean after the call, and the
    digests of the parameter gradients."""
    from oracle import seeded
    nd = _import_ref_networks()
    from torch_utils import training_stats
    from training.loss import StyleGAN2Loss
    import training.loss as loss_mod
    B, bg, seed = 4, 64, 13
    inp = seeded.comp_inputs(B, bg, seed)
    G, D = _ref_G_D(nd, bg, inp['feats_g'], inp['feats_d'])
    skip = ('backbone.0.body.', 'text_encoder.', 'text_decoder.')
    G.load_state_dict(seeded.seeded_state_dict(G, 1, skip)); D.load_state_dict(seeded.seeded_state_dict(D, 2, skip))
    d = {'B': np.asarray(B), 'bg': np.asarray(bg), 'seed': np.asarray(seed), 'r1_gamma': np.asarray(10.0), 'pl_weight': np.asarray(2.0)}
    patch = torch.zeros(B, 9, 1, 1, 1)
    c = torch.zeros(B, 0)
    tok = G.tokenizer(sum(inp['texts'], []))
    d['text_feat'] = G.text_encoder(tok.input_ids, tok.attention_mask).last_hidden_state[:, 0, :].view(B, 9, -1)
    d['text_len'] = torch.tensor([len(t) for t in sum(inp['texts'], [])]).view(B, 9)
    reports = {}
    real_report = training_stats.report
    training_stats.report = lambda name, value: reports.setdefault(name, []).append(torch.as_tensor(value).detach().clone()) or value
    loss_mod.training_stats.report = training_stats.report
    real_randn_like = torch.randn_like
    for tag, dt in (('', torch.float32), ('64', torch.float64)):
        G.to(dt); D.to(dt)
        D.bg_decoder.float()
        cast = lambda t: t.to(dt) if t.dtype.is_floating_point else t
        loss = StyleGAN2Loss(torch.device('cpu'), G, D, r1_gamma=10.0, pl_weight=2.0, pl_batch_shrink=2)
        loss.pl_mean = loss.pl_mean.to(dt)
        G.requires_grad_(False); D.requires_grad_(False)
        grads = {}
        for phase, mod_ in (('Greg', G), ('Dboth', D), ('Dmain', D)):
            reports.clear()
            mod_.requires_grad_(True); mod_.text_encoder.requires_grad_(False)
            for p in mod_.parameters():
                p.grad = None
            noise = {}

            def randn_like(t, *a, **k):
                torch.manual_seed(77)
                r = real_randn_like(t.float(), *a, **k).to(t.dtype)      # the same draw in both precisions
                noise['v'] = r.detach().clone()
                return r
            torch.randn_like = randn_like
            try:
                loss.accumulate_gradients(phase=phase, bbox_real=cast(inp['bbox_real']), bbox_class=inp['bbox_class'], bbox_text=inp['texts'], bbox_patch=cast(patch),
                                          padding_mask=inp['padding_mask'], background=inp['background'], real_c=cast(c),
                                          gen_z=cast(inp['z_g'] if phase == 'Greg' else inp['z_d']), gen_c=cast(c), gain=4 if phase == 'Greg' else 1, cur_nimg=0)
            finally:
                torch.randn_like = real_randn_like
            mod_.requires_grad_(False)
            grads[phase] = {name: p.grad.detach().clone() for name, p in mod_.named_parameters() if p.grad is not None}
            if phase == 'Greg':
                d['pl_noise'] = noise['v'].float()
                d[f'Greg/pl_mean{tag}'] = loss.pl_mean.detach().clone()
            if phase != 'Dmain':
                for name, vals in reports.items():
                    for i, v in enumerate(vals):
                        d[f'{phase}/report{tag}/{name}' + (f'#{i}' if len(vals) > 1 else '')] = v
        for name, g in grads['Greg'].items():
            st, sub = seeded.grad_digest(g)
            d[f'Greg/gstat{tag}/{name}'] = st; d[f'Greg/gsub{tag}/{name}'] = sub if tag else sub.astype(np.float32)
        for name, g in grads['Dboth'].items():
            st, sub = seeded.grad_digest(g - grads['Dmain'][name])
            d[f'R1/gstat{tag}/{name}'] = st; d[f'R1/gsub{tag}/{name}'] = sub if tag else sub.astype(np.float32)
        if not tag:
            d['Greg/params_with_grad'] = np.asarray(sorted(grads['Greg']))
    training_stats.report = real_report
    save('reg', d)


def gen_config0():
    """BASELINE.json configs[0]: ONE sample, 128x128 background (stride 32 -> 4x4 = 16 memory tokens), 3 text boxes valid of the 9 slots,
    the reference's own `Generator.forward` (networks_detr.py:133-187) on CPU, reconst off and on.  Same stand-ins as gen_composition
    (ResNet body = the seeded feature map, tokenizer / text encoder table, LM decoder zero loss); weights and inputs from oracle/seeded.py."""
    from oracle import seeded
    nd = _import_ref_networks()
    B, bg, seed, nvalid = 1, 128, 23, 3
    inp = seeded.comp_inputs(B, bg, seed)
    pm = torch.ones(B, 9, dtype=torch.bool); pm[:, :nvalid] = False
    G, _ = _ref_G_D(nd, bg, inp['feats_g'], inp['feats_d'])
    skip = ('backbone.0.body.', 'text_encoder.', 'text_decoder.')
    G.load_state_dict(seeded.seeded_state_dict(G, 1, skip))
    d = {'B': np.asarray(B), 'bg': np.asarray(bg), 'seed': np.asarray(seed), 'nvalid': np.asarray(nvalid)}
    patch = torch.zeros(B, 9, 1, 1, 1); c = torch.zeros(B, 0)
    tok = G.tokenizer(sum(inp['texts'], []))
    d['text_feat'] = G.text_encoder(tok.input_ids, tok.attention_mask).last_hidden_state[:, 0, :].view(B, 9, -1)
    d['text_len'] = torch.tensor([len(t) for t in sum(inp['texts'], [])]).view(B, 9)
    with torch.no_grad():
        d['G/bbox_fake_noreconst'] = G(inp['z_g'], inp['bbox_class'], inp['bbox_real'], inp['texts'], patch, pm, inp['background'], c)
        out = G(inp['z_g'], inp['bbox_class'], inp['bbox_real'], inp['texts'], patch, pm, inp['background'], c, True)
        for k, v in zip(('bbox_fake', 'loss_z', 'logit_cls', 'loss_lm', 'loss_text_len'), out):
            d['G/' + k] = v
    assert d['G/logit_cls'].shape[0] == nvalid
    save('config0', d)


def gen_box_ops():
    """north_star row ns-1: the reference's own detr_util/box_ops.py (box_cxcywh_to_xyxy :19-23, box_iou :35-48, generalized_box_iou
    :51-71; torchvision.ops.boxes.box_area is the two-line area formula, stubbed above) on batches of layouts, and the Hungarian
    assignment the DETR matcher derives from it: scipy.optimize.linear_sum_assignment on cost = -GIoU (ties: duplicated boxes,
    prediction == target, quantised coordinates; a zero-area target)."""
    from scipy.optimize import linear_sum_assignment
    from detr_util import box_ops
    g = torch.Generator().manual_seed(77)
    d = {}
    idx = 0
    for n in (9, 9, 9, 9, 9, 9, 16, 3, 1):
        pred = torch.cat([torch.rand(n, 2, generator=g) * 0.6 + 0.2, torch.rand(n, 2, generator=g) * 0.35 + 0.05], -1)
        tgt = torch.cat([torch.rand(n, 2, generator=g) * 0.6 + 0.2, torch.rand(n, 2, generator=g) * 0.35 + 0.05], -1)
        if idx == 1:
            tgt = pred.clone()                               # identical layouts
        if idx == 2 and n > 4:
            tgt[3] = tgt[1]; pred[5] = pred[2]; pred[6] = pred[2]    # duplicated boxes -> tied rows / columns
        if idx == 3:
            pred = (pred * 8).round() / 8; tgt = (tgt * 8).round() / 8; pred[:, 2:].clamp_(min=0.125); tgt[:, 2:].clamp_(min=0.125)   # quantised
        if idx == 4:
            tgt[2, 2:] = 0.0                                 # zero-area target box (union > 0 against every prediction)
        if idx == 5:
            pred[:] = pred[0]                                # constant rows
        p_xyxy, t_xyxy = box_ops.box_cxcywh_to_xyxy(pred), box_ops.box_cxcywh_to_xyxy(tgt)
        iou, union = box_ops.box_iou(p_xyxy, t_xyxy)
        giou = box_ops.generalized_box_iou(p_xyxy, t_xyxy)
        cost = (-giou).double().numpy()
        r, c = linear_sum_assignment(cost)
        for k, v in (('pred', pred), ('tgt', tgt), ('p_xyxy', p_xyxy), ('iou', iou), ('union', union), ('giou', giou)):
            d[f'{k}{idx}'] = v
        d[f'back{idx}'] = box_ops.box_xyxy_to_cxcywh(p_xyxy)
        d[f'row{idx}'] = r; d[f'col{idx}'] = c
        idx += 1
    # rectangular pairwise matrices (N != M)
    a = torch.cat([torch.rand(5, 2, generator=g) * 0.5, torch.rand(5, 2, generator=g) * 0.4 + 0.5], -1)
    b = torch.cat([torch.rand(9, 2, generator=g) * 0.5, torch.rand(9, 2, generator=g) * 0.4 + 0.5], -1)
    d['rect_a'] = a; d['rect_b'] = b
    d['rect_iou'], d['rect_union'] = box_ops.box_iou(a, b)
    d['rect_giou'] = box_ops.generalized_box_iou(a, b)
    d['count'] = np.asarray(idx)
    save('box_ops', d)


if __name__ == '__main__':
    sys.path.insert(0, REF)
    sys.path.insert(1, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # `oracle.seeded` (the reference has no `oracle` package)
    _stub_modules()
    torch.set_num_threads(8)
    if '--only-composition' in sys.argv:
        gen_composition(); gen_config0(); sys.exit(0)
    if '--only-reg' in sys.argv:
        gen_reg(); sys.exit(0)
    if '--only-config0' in sys.argv:
        gen_config0(); sys.exit(0)
    if '--only-box-ops' in sys.argv:
        gen_box_ops(); sys.exit(0)
    if '--only-metrics' in sys.argv:
        gen_metrics(); sys.exit(0)
    if '--only-resample' in sys.argv:
        gen_resample(); sys.exit(0)
    if '--only-bert' in sys.argv:
        gen_bert(); gen_bert_lm(); sys.exit(0)
    if '--skip-done' not in sys.argv:
        gen_ops(); gen_transformer()
    gen_losses(); gen_decoder(); gen_frozen_bn(); gen_lsap(); gen_dp_step(); gen_metrics(); gen_bert(); gen_bert_lm(); gen_resample(); gen_composition(); gen_reg(); gen_config0(); gen_box_ops()
