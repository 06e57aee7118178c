"""GPU parity of the module-level hot path (seam 1) against (a) golden vectors captured from the reference
and (b) the CPU oracle on seeded inputs.  Tolerance: 1e-3 relative fp32 on outputs / losses / bbox
(north_star); most checks are far tighter and say so."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

G_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load(name):
    d = np.load(os.path.join(G_DIR, name + '.npz'), allow_pickle=False)
    return {k: torch.from_numpy(d[k]) for k in d.files if d[k].ndim > 0 or d[k].dtype.kind in 'fiub'}


def sd_of(d, prefix='sd/'):
    return {k[len(prefix):]: v for k, v in d.items() if k.startswith(prefix)}


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


def check(a, b, tol, what):
    e = rel(a, b)
    assert e <= tol, f'{what}: rel err {e:.3e} > {tol:.1e}'


def same_up_to_relu_flips(a, b, what):
    """Two fp32 evaluations of a relu network that sum in different orders: one hidden unit among millions may land on the other side of its
    relu.  That rewrites ONE row of that layer's weight gradients (O(1) relative to the row, a few % of the tensor's largest entry) and
    moves every tensor upstream of it densely by ~1e-3.  A wiring error is dense AND large.  Accept: every entry within 2e-2 of the
    tensor's largest, or at most 1 % of the entries beyond that."""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    e = (a - b).abs() / (b.abs().max() + 1e-12)
    frac = (e > 2e-2).double().mean().item()
    assert e.max().item() <= 2e-2 or frac <= 0.01, f'{what}: max {e.max().item():.3e}, {frac:.4f} of the entries beyond 2e-2'
    # ... and a dense criterion beside the outlier allowance: a wiring error confined to a slice (one bias, one block of in_proj rows) or a
    # ~1 % scale error moves the bulk of the entries (median ~3e-3 of the largest entry); a flipped unit does not, and fp32 summation-order noise of a
    # bias gradient (a column sum over all tokens) measures 1.6e-4.  Median <= 5e-4, 90th percentile <= 5e-3 of the largest entry, for the tensor
    # and for each third of its leading dimension (the q / k / v row blocks of a packed projection) on the block's own scale.
    def dense(ee, tag, med_tol=5e-4, p90_tol=5e-3):
        v = ee.reshape(-1)
        if v.numel() == 0:
            return
        med, p90 = v.median().item(), v.kthvalue(max(1, int(0.9 * v.numel()))).values.item()
        assert med <= med_tol and p90 <= p90_tol, f'{what}{tag}: median error {med:.3e}, p90 {p90:.3e} of the largest entry'
    dense(e, '')
    if e.dim() >= 1 and e.shape[0] >= 3 and e.shape[0] % 3 == 0:
        t = e.shape[0] // 3
        for i in range(3):
            blk, ref = e[i * t:(i + 1) * t], b[i * t:(i + 1) * t].abs().max()
            if ref > 1e-3 * b.abs().max():      # relative to the block's own scale when the block is not negligible
                dense(blk * (b.abs().max() / ref), f' rows {i * t}:{(i + 1) * t}')
    return e.max().item()


@pytest.mark.parametrize('name', ['transformer', 'transformer_token'])
def test_detr_transformer_matches_reference_golden(dev, name):
    from layoutdetr_amd.training.detr_transformer import Transformer, TransformerWithToken
    d = load(name)
    cls = TransformerWithToken if name.endswith('token') else Transformer
    m = cls(d_model=64, nhead=2, num_encoder_layers=2, num_decoder_layers=2, dim_feedforward=128, dropout=0.1).eval()
    m.load_state_dict(sd_of(d))
    m.to(dev)
    src = d['src'].to(dev).requires_grad_(True); tgt = d['tgt'].to(dev).requires_grad_(True)
    hs, mem = m(src, d['mask'].to(dev), d['pos'].to(dev), tgt, d['kpm'].to(dev))
    check(hs, d['hs'], 2e-5, 'hs'); check(mem, d['mem'], 2e-5, 'memory')
    ((hs * d['g_hs'].to(dev)).sum() + (mem * d['g_mem'].to(dev)).sum()).backward()
    check(src.grad, d['d_src'], 1e-4, 'd_src'); check(tgt.grad, d['d_tgt'], 1e-4, 'd_tgt')
    grads = sd_of(d, 'grad/')
    for k, p in m.named_parameters():
        if k in grads:
            check(p.grad, grads[k], 2e-4, 'grad ' + k)


@pytest.mark.parametrize('flat', [False, True])
def test_grouped_memory_kv_projection_equals_per_layer_projections(dev, flat):
    """The six decoder layers' memory K / V projections as two grouped GEMMs (hip.attention._GroupedKVFn; backward: the layers' attention
    kernels write dK / dV into one buffer, two data-gradient + two weight-gradient GEMMs, one strided add into the flat .grad) against the
    per-layer launches the reference's call structure maps to (detr_transformer.py:277-280), at the real sizes (d 256, 8 heads, 6 + 6
    layers, 16 samples x 64 memory tokens): outputs, input gradients and every parameter gradient, with autograd-returned gradients
    (flat=False) and with gradients accumulated into a FlatModule buffer on top of a non-zero previous content (flat=True)."""
    from layoutdetr_amd.training import detr_transformer as T
    from layoutdetr_amd.training.training_loop import FlatModule
    torch.manual_seed(91)
    B, S, N, d = 16, 64, 9, 256
    m = T.TransformerWithToken(d_model=d, nhead=8, num_encoder_layers=6, num_decoder_layers=6, dim_feedforward=2048, dropout=0.1).eval().to(dev)
    src0 = torch.randn(B, d, 8, 8, device=dev); pos = torch.randn(B, d, 8, 8, device=dev) * 0.3
    mask = torch.zeros(B, 8, 8, dtype=torch.bool, device=dev); mask[1, :, 6:] = True; mask[5, 5:, :] = True
    tgt0 = torch.randn(N, B, d, device=dev); kpm = torch.zeros(B, N, dtype=torch.bool, device=dev); kpm[2, 4:] = True; kpm[7, 1:] = True
    g_hs = torch.randn(B, N + 1, d, device=dev); g_mem = torch.randn(B, d, 8, 8, device=dev) * 0.1
    fm = FlatModule(m) if flat else None
    res = {}
    prev = T._GROUP_KV
    try:
        for grouped in (False, True):
            T._GROUP_KV = grouped
            if flat:
                fm.zero_grad(); fm.gflat.fill_(0.25)          # accumulate on top of existing content
            else:
                for p_ in m.parameters():
                    p_.grad = None
            src = src0.clone().requires_grad_(True); tgt = tgt0.clone().requires_grad_(True)
            hs, mem = m(src, mask, pos, tgt, kpm)
            ((hs * g_hs).sum() + (mem * g_mem).sum()).backward()
            res[grouped] = dict(hs=hs.detach().clone(), mem=mem.detach().clone(), d_src=src.grad.clone(), d_tgt=tgt.grad.clone(),
                                **{'g/' + k: p_.grad.detach().clone() for k, p_ in m.named_parameters() if p_.grad is not None})
    finally:
        T._GROUP_KV = prev
    assert set(res[True]) == set(res[False])
    worst = 0.0
    for k, v in res[False].items():
        e = rel(res[True][k], v)
        worst = max(worst, e)
        if k in ('hs', 'mem'):
            assert e <= 2e-5, f'{k}: {e:.3e}'
        else:
            same_up_to_relu_flips(res[True][k], v, k)      # (a relu unit flipped by the different rounding of K / V; see the helper)
    assert any('multihead_attn.in_proj_weight' in k for k in res[True])
    print(f'[grouped K/V flat={flat}] worst deviation from the per-layer path {worst:.2e}')


@pytest.mark.parametrize('flat', [False, True])
def test_fused_ffn_stacks_equal_unfused_stacks(dev, flat):
    """The decoder-side stacks with the fused feed-forward block (hip/ffn.py: 2 launches per direction) against the same stacks on the
    two-GEMM path: TransformerWithToken (6 + 6 layers, the decoder sees 10 tokens x 16 samples) and a 6-layer token encoder (9 x 16),
    outputs, input gradients and every parameter gradient; gradients returned by autograd (flat=False) and accumulated into a
    FlatModule buffer (flat=True: fp32 atomics straight into the flat .grad)."""
    from layoutdetr_amd.hip import ffn as hffn
    from layoutdetr_amd.training import detr_transformer as T
    from layoutdetr_amd.training.training_loop import FlatModule
    torch.manual_seed(92)
    B, N, d = 16, 9, 256
    m = T.TransformerWithToken(d_model=d, nhead=8, num_encoder_layers=2, num_decoder_layers=6, dim_feedforward=2048, dropout=0.1).eval().to(dev)
    enc = T.TransformerEncoder(T.TransformerEncoderLayer(d_model=d, nhead=8, dim_feedforward=2048), num_layers=6).eval().to(dev)
    both = torch.nn.ModuleList([m, enc])
    src0 = torch.randn(B, d, 2, 2, device=dev); pos = torch.randn(B, d, 2, 2, device=dev) * 0.3      # 4 memory tokens: the encoder runs fused as well
    mask = torch.zeros(B, 2, 2, dtype=torch.bool, device=dev); mask[3, :, 1:] = True
    tgt0 = torch.randn(N, B, d, device=dev); kpm = torch.zeros(B, N, dtype=torch.bool, device=dev); kpm[2, 4:] = True; kpm[7, 1:] = True
    x0 = torch.randn(B * N, d, device=dev)
    g_hs = torch.randn(B, N + 1, d, device=dev); g_enc = torch.randn(B * N, d, device=dev)
    fm = FlatModule(both) if flat else None
    res = {}
    prev = hffn.FUSED
    try:
        for fused in (False, True):
            hffn.FUSED = fused
            if flat:
                fm.zero_grad(); fm.gflat.fill_(0.125)
            else:
                for p_ in both.parameters():
                    p_.grad = None
            src = src0.clone().requires_grad_(True); tgt = tgt0.clone().requires_grad_(True); x = x0.clone().requires_grad_(True)
            hs, _ = m(src, mask, pos, tgt, kpm)
            y = enc.forward2d(x, B, N, kpm, None)
            ((hs * g_hs).sum() + (y * g_enc).sum()).backward()
            res[fused] = dict(hs=hs.detach().clone(), y=y.detach().clone(), d_src=src.grad.clone(), d_tgt=tgt.grad.clone(), d_x=x.grad.clone(),
                              **{'g/' + k: p_.grad.detach().clone() for k, p_ in both.named_parameters() if p_.grad is not None})
    finally:
        hffn.FUSED = prev
    assert set(res[True]) == set(res[False])
    worst = 0.0
    for k, v in res[False].items():
        e = rel(res[True][k], v)
        worst = max(worst, e)
        # forward values to 2e-5; gradients: the two paths sum the hidden pre-activations in different orders, so among millions of hidden
        # units one can land on the other side of its relu (seen: 2.6e-3 on every tensor upstream of it) -- a wiring error (missing bias
        # gradient, wrong dropout scale, a lost accumulation into the flat buffer) is O(1); the kernel itself is held to 2e-5 against fp64
        # entry by entry in test_kernels_gpu.py::test_ffn_fused_block_vs_fp64_reference
        if k in ('hs', 'y'):
            assert e <= 2e-5, f'{k}: {e:.3e}'
        else:
            same_up_to_relu_flips(res[True][k], v, k)
    print(f'[fused FFN flat={flat}] worst deviation from the two-GEMM path {worst:.2e}')


@pytest.mark.parametrize('flat,train', [(False, False), (True, False), (True, True)])
def test_fused_self_attention_stacks_equal_unfused_stacks(dev, flat, train):
    """The <= 16-token stacks as ONE autograd node per stack (hip/stacks.py: group launches of csrc/mha_small.hip / ffn_fused.hip / layernorm.hip,
    gradients between sub-blocks as partial sums, attention backward and the layer's weight gradients as one launch each) against the
    per-sub-block path on the generic kernels (stacks.ENABLED = False): TransformerWithToken's decoder (10 tokens x 16 samples) and a 6-layer
    token encoder (9 x 16, ragged key-padding masks): outputs, input gradients and every parameter gradient.  train=True: both paths draw their
    dropout seeds in the same order and index the attention mask identically, so they must still agree."""
    from layoutdetr_amd.hip import stacks as A
    from layoutdetr_amd.hip import core
    from layoutdetr_amd.training import detr_transformer as T
    from layoutdetr_amd.training.training_loop import FlatModule
    torch.manual_seed(93)
    B, N, d = 16, 9, 256
    m = T.TransformerWithToken(d_model=d, nhead=8, num_encoder_layers=1, num_decoder_layers=6, dim_feedforward=2048, dropout=0.1).to(dev)
    enc = T.TransformerEncoder(T.TransformerEncoderLayer(d_model=d, nhead=8, dim_feedforward=2048), num_layers=6).to(dev)
    both = torch.nn.ModuleList([m, enc]).train(train)
    src0 = torch.randn(B, d, 2, 2, device=dev); pos = torch.randn(B, d, 2, 2, device=dev) * 0.3
    mask = torch.zeros(B, 2, 2, dtype=torch.bool, device=dev); mask[3, :, 1:] = True
    tgt0 = torch.randn(N, B, d, device=dev); kpm = torch.zeros(B, N, dtype=torch.bool, device=dev); kpm[2, 4:] = True; kpm[7, 1:] = True
    x0 = torch.randn(B * N, d, device=dev)
    g_hs = torch.randn(B, N + 1, d, device=dev); g_enc = torch.randn(B * N, d, device=dev)
    fm = FlatModule(both) if flat else None
    res = {}
    prev = A.ENABLED
    try:
        for fused in (False, True):
            A.ENABLED = fused
            core._seed_counter[0] = 1000
            if flat:
                fm.zero_grad(); fm.gflat.fill_(0.125)
            else:
                for p_ in both.parameters():
                    p_.grad = None
            src = src0.clone().requires_grad_(True); tgt = tgt0.clone().requires_grad_(True); x = x0.clone().requires_grad_(True)
            n0 = core.engine_launch_counts() if hasattr(core, 'engine_launch_counts') else None
            hs, _ = m(src, mask, pos, tgt, kpm)
            y = enc.forward2d(x, B, N, kpm, None)
            ((hs * g_hs).sum() + (y * g_enc).sum()).backward()
            res[fused] = dict(hs=hs.detach().clone(), y=y.detach().clone(), d_src=src.grad.clone(), d_tgt=tgt.grad.clone(), d_x=x.grad.clone(),
                              **{'g/' + k: p_.grad.detach().clone() for k, p_ in both.named_parameters() if p_.grad is not None})
    finally:
        A.ENABLED = prev
    assert set(res[True]) == set(res[False])
    assert any('self_attn.in_proj_weight' in k for k in res[True]) and any('self_attn.out_proj.bias' in k for k in res[True])
    worst = 0.0
    for k, v in res[False].items():
        e = rel(res[True][k], v)
        worst = max(worst, e)
        if k in ('hs', 'y'):
            assert e <= (2e-5 if not train else 1e-4), f'{k}: {e:.3e}'
        else:
            same_up_to_relu_flips(res[True][k], v, k)
    print(f'[fused self-attention flat={flat} train={train}] worst deviation from the three-launch path {worst:.2e}')


@pytest.mark.parametrize('cls_name,d_model,nhead', [('Transformer', 256, 4), ('TransformerWithToken', 256, 2), ('Transformer', 192, 2)])
def test_detr_transformer_wider_heads_vs_oracle(dev, cls_name, d_model, nhead):
    """Head widths 64 / 128 / 96 (the reference's constructors take any hidden_dim / nhead; its own default is 256 / 8 = 32): DETR
    encoder-decoder forward and every gradient against the golden-pinned oracle (oracle/detr_ref.transformer) on ragged masks."""
    from layoutdetr_amd.training import detr_transformer as T
    from oracle import detr_ref
    torch.manual_seed(71)
    B, Hh, Ww, N = 3, 4, 5, 9
    m = getattr(T, cls_name)(d_model=d_model, nhead=nhead, num_encoder_layers=2, num_decoder_layers=2, dim_feedforward=256, dropout=0.1).eval()
    for p in m.parameters():
        if p.dim() == 1:
            p.data.uniform_(-0.2, 0.2).add_(1.0 if p.numel() == d_model and p.data.mean() > 0.5 else 0.0)
    src = torch.randn(B, d_model, Hh, Ww); pos = torch.randn(B, d_model, Hh, Ww) * 0.5
    mask = torch.zeros(B, Hh, Ww, dtype=torch.bool); mask[1, :, 3:] = True; mask[2, 2:, :] = True
    tgt = torch.randn(N, B, d_model); kpm = torch.zeros(B, N, dtype=torch.bool); kpm[0, 6:] = True; kpm[2, 1:] = True
    with_token = cls_name.endswith('Token')
    valid = torch.cat([torch.ones(B, 1, dtype=torch.bool), ~kpm], 1) if with_token else ~kpm          # [B, Lq(+1)]
    g_hs = torch.randn(B, valid.shape[1], d_model) * valid[..., None]
    sd = {k: v.clone().requires_grad_(v.dtype.is_floating_point) for k, v in m.state_dict().items()}
    sr, tr = src.clone().requires_grad_(True), tgt.clone().requires_grad_(True)
    hs_r, mem_r = detr_ref.transformer(sd, sr, mask, pos, tr, kpm, nhead, with_token=with_token)
    (hs_r * g_hs).sum().backward()
    m.to(dev)
    sg, tg = src.to(dev).requires_grad_(True), tgt.to(dev).requires_grad_(True)
    hs, mem = m(sg, mask.to(dev), pos.to(dev), tg, kpm.to(dev))
    check(hs.reshape(hs_r.shape)[valid.to(dev)], hs_r[valid], 3e-5, 'hs')
    (hs.reshape(hs_r.shape) * g_hs.to(dev)).sum().backward()
    check(sg.grad, sr.grad, 2e-4, 'd_src'); check(tg.grad.transpose(0, 1)[(~kpm).to(dev)], tr.grad.transpose(0, 1)[~kpm], 2e-4, 'd_tgt')
    n = 0
    for k, p in m.named_parameters():
        if sd[k].grad is not None and sd[k].grad.abs().max() > 0:
            check(p.grad, sd[k].grad, 3e-4, 'grad ' + k); n += 1
    assert n >= 40


def test_layoutganpp_encoder_matches_reference_golden(dev):
    from layoutdetr_amd.training.util import TransformerWithToken_layoutganpp
    d = load('transformer_layoutganpp')
    m = TransformerWithToken_layoutganpp(d_model=64, nhead=2, dim_feedforward=128, num_layers=2).eval()
    m.load_state_dict(sd_of(d)); m.to(dev)
    x = d['x'].to(dev).requires_grad_(True)
    y = m(x, src_key_padding_mask=d['kpm'].to(dev))
    valid = torch.cat([torch.ones(2, 1, dtype=torch.bool), ~d['kpm']], 1).t()  # [L+1, B]; padded rows are never compared
    check(y[valid.to(dev)], d['y'][valid], 2e-5, 'y')
    # the fixture's backward used the full upstream gradient g (padded rows included).  Padded rows only feed themselves (they are
    # masked out as keys), so zeroing their upstream gradient on BOTH sides isolates the defined part: the reference value of that
    # is recomputed by the pinned CPU oracle (tests/test_oracle_golden.py holds it to the fixture's full-g gradients).
    from oracle import detr_ref
    g = d['g'].clone(); g[~valid] = 0
    (y * g.to(dev)).sum().backward()
    sd = {k: v.clone().requires_grad_(v.dtype.is_floating_point) for k, v in sd_of(d).items()}
    xr = d['x'].clone().requires_grad_(True)
    yr = detr_ref.token_encoder_layoutganpp(sd, '', xr, d['kpm'], 2)
    check(yr[valid], d['y'][valid], 2e-5, 'oracle y')
    (yr * g).sum().backward()
    check(x.grad, xr.grad, 1e-4, 'd_x')
    n_checked = 0
    for k, p in m.named_parameters():
        if sd[k].grad is not None and sd[k].grad.abs().max() > 0:
            check(p.grad, sd[k].grad, 2e-4, 'grad ' + k); n_checked += 1
    assert n_checked >= 20


def test_stylegan2_decoder_matches_reference_golden(dev):
    from layoutdetr_amd.training.networks_stylegan2 import Decoder
    d = load('decoder')
    m = Decoder(z_dim=32, w_dim=32, img_resolution=16, img_channels=3, use_noise=False, channel_base=256, channel_max=32,
                num_fp16_res=0, conv_clamp=None, fused_modconv_default=False).train()
    m.load_state_dict(sd_of(d)); m.to(dev)
    z = d['z'].to(dev).requires_grad_(True)
    img = m(z)
    assert img.shape == (2, 3, 16, 16)
    check(img, d['img'], 2e-5, 'img')
    (img * d['g'].to(dev)).sum().backward()
    check(z.grad, d['d_z'], 2e-4, 'd_z')
    grads = sd_of(d, 'grad/')
    for k, p in m.named_parameters():
        check(p.grad, grads[k], 5e-4, 'grad ' + k)


def test_position_encoding_and_frozen_bn(dev):
    from layoutdetr_amd.detr_util.misc import NestedTensor
    from layoutdetr_amd.training.detr_backbone import FrozenBatchNorm2d
    from layoutdetr_amd.training.detr_position_encoding import PositionEmbeddingSine
    d = load('pos_encoding')
    pe = PositionEmbeddingSine(128, normalize=True)
    pos = pe(NestedTensor(torch.zeros(2, 1, 4, 5, device=dev), d['mask'].to(dev)))
    check(pos, d['pos'], 1e-6, 'pos')
    d = load('frozen_bn')
    bn = FrozenBatchNorm2d(6)
    bn.load_state_dict({k: d[k] for k in ('weight', 'bias', 'running_mean', 'running_var')})
    check(bn.to(dev)(d['x'].to(dev)), d['y'], 1e-6, 'frozen bn')


def _make(dev, bg=64, seed=0):
    from layoutdetr_amd.training.networks_detr import Discriminator, Generator
    torch.manual_seed(seed)
    kw = dict(num_bbox_labels=8, img_channels=3, img_height=bg, img_width=bg, c_dim=0, background_size=bg, bert_f_dim=768, im_f_dim=512)
    G = Generator(z_dim=4, **kw)
    D = Discriminator(**kw)
    # non-trivial frozen-BN statistics so the folded affine is exercised
    for m in list(G.modules()) + list(D.modules()):
        if m.__class__.__name__ == 'FrozenBatchNorm2d':
            m.weight.uniform_(0.5, 1.5); m.bias.normal_(0, 0.1); m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5)
    return G, D


def _batch(B, bg, seed=1):
    g = torch.Generator().manual_seed(seed)
    bt = dict(bbox_real=torch.cat([torch.rand(B, 9, 2, generator=g) * 0.6 + 0.2, torch.rand(B, 9, 2, generator=g) * 0.35 + 0.05], -1),
              bbox_class=torch.randint(0, 8, (B, 9), generator=g), text_feat=torch.randn(B, 9, 768, generator=g),
              text_len=torch.randint(1, 40, (B, 9), generator=g), padding_mask=torch.zeros(B, 9, dtype=torch.bool),
              background=torch.randn(B, 3, bg, bg, generator=g))
    bt['padding_mask'][0, 6:] = True
    return bt, torch.randn(B, 9, 4, generator=g), torch.randn(B, 9, 4, generator=g)


def test_generator_discriminator_forward_vs_oracle(dev):
    from layoutdetr_amd.training.networks_detr import TextFeatures
    from oracle import networks_ref
    G, D = _make(dev)
    bt, z, _ = _batch(2, 64)
    Gsd = {k: v.clone() for k, v in G.state_dict().items()}; Dsd = {k: v.clone() for k, v in D.state_dict().items()}
    with torch.no_grad():
        ref_g = networks_ref.generator(Gsd, z, bt['bbox_class'], bt['text_feat'], bt['text_len'], bt['padding_mask'], bt['background'], reconst=True)
        ref_d = networks_ref.discriminator(Dsd, bt['bbox_real'], bt['bbox_class'], bt['text_feat'], bt['text_len'], bt['padding_mask'],
                                           bt['background'], reconst=True, bg_size=64)
    G.eval().to(dev); D.eval().to(dev)
    tf = TextFeatures(bt['text_feat'].to(dev), bt['text_len'].to(dev))
    patch = torch.zeros(2, 9, 1, 1, 1, device=dev)
    with torch.no_grad():
        out_g = G(z.to(dev), bt['bbox_class'].to(dev), bt['bbox_real'].to(dev), tf, patch, bt['padding_mask'].to(dev), bt['background'].to(dev), None, reconst=True)
        out_d = D(bt['bbox_real'].to(dev), bt['bbox_class'].to(dev), tf, patch, bt['padding_mask'].to(dev), bt['background'].to(dev), None, reconst=True)
    valid = ~bt['padding_mask']
    check(out_g[0][valid.to(dev)], ref_g[0][valid], 1e-3, 'bbox_fake')
    for i, nm in [(1, 'loss_z'), (2, 'logit_cls'), (4, 'loss_text_len')]:
        check(out_g[i], ref_g[i], 1e-3, 'G ' + nm)
    names = ['logit', 'logit_uncond', 'bbox_pred', 'logit_cls', 'loss_lm', 'loss_text_len', 'bg_rec', 'bbox_pred_uncond', 'logit_cls_uncond']
    for i, nm in enumerate(names):
        if nm != 'loss_lm':
            check(out_d[i], ref_d[i], 1e-3, 'D ' + nm)


def test_training_iteration_vs_oracle(dev):
    """Full Gmain + Dmain iteration (fwd, bwd, DP post-processing, Adam) on the HIP path vs the CPU oracle."""
    from layoutdetr_amd.training import training_loop as tl
    from layoutdetr_amd.training.loss import StyleGAN2Loss
    from layoutdetr_amd.training.networks_detr import TextFeatures
    from oracle import step_ref
    G, D = _make(dev, seed=3)
    bt, zg, zd = _batch(2, 64, seed=4)
    Gsd = {k: v.clone() for k, v in G.state_dict().items()}; Dsd = {k: v.clone() for k, v in D.state_dict().items()}
    out, gG, gD, Gn, Dn = step_ref.training_iteration(Gsd, Dsd, bt, zg, zd, lr=1e-5, bg_size=64,
                                                      G_param_names={n for n, _ in G.named_parameters()},
                                                      D_param_names={n for n, _ in D.named_parameters()})
    G.eval().requires_grad_(False).to(dev); D.eval().requires_grad_(False).to(dev)   # eval: dropout off (parity mode)
    pG = tl.Phase('Gmain', G, lr=1e-5, reg_interval=None); pD = tl.Phase('Dmain', D, lr=1e-5, reg_interval=None)
    loss = StyleGAN2Loss(dev, G, D)
    dp = tl.DataParallelStep(world_size=1, fuse_sanitize=True)
    batch = dict(bbox_real=bt['bbox_real'].to(dev), bbox_class=bt['bbox_class'].to(dev),
                 bbox_text=TextFeatures(bt['text_feat'].to(dev), bt['text_len'].to(dev)), bbox_patch=torch.zeros(2, 9, 1, 1, 1, device=dev),
                 padding_mask=bt['padding_mask'].to(dev), background=bt['background'].to(dev), real_c=torch.zeros(2, 0, device=dev),
                 gen_c=torch.zeros(2, 0, device=dev))
    # snapshot gradients of each phase before Adam consumes them
    grads = {}
    orig_apply = dp.apply

    def spy(phase, **kw):
        grads[phase.name] = {n: p.grad.detach().clone() for n, p in phase.module.named_parameters()}
        orig_apply(phase, **kw)
    dp.apply = spy
    tl.training_iteration(loss, [pG, pD], dp, batch, 2, [zg.to(dev), zd.to(dev)])
    check(loss.last['bbox_fake'][(~bt['padding_mask']).to(dev)], out['bbox_fake'][~bt['padding_mask']], 1e-3, 'bbox_fake')
    # Gradients: the step is only piecewise differentiable (ReLU / max-pool).  A single ReLU whose pre-activation is
    # ~1e-5 on one side and 0 on the other flips its mask and moves a few weight-gradient rows by percents while every
    # value still agrees to 1e-6 (measured: 1 flip in 32768 activations of layer3.0 -> 3e-2 on that conv's dW).  So the
    # gate is distributional: median and 90th percentile over all parameter tensors tight, maximum bounded.
    errs = []
    for name, ref in (('Gmain', gG), ('Dmain', gD)):
        errs += [rel(grads[name][k], g) for k, g in ref.items()]
    errs = torch.tensor(errs)
    worst = errs.max().item()
    assert errs.median().item() <= 1e-4, f'median grad rel err {errs.median().item():.3e}'
    assert errs.quantile(0.9).item() <= 5e-3, f'p90 grad rel err {errs.quantile(0.9).item():.3e}'
    assert worst <= 0.15, f'worst grad rel err {worst:.3e}'
    # Adam: parameters moved by ~lr in the direction of -sign(g) where |g| is not tiny
    for k, p in G.named_parameters():
        if k in gG:
            mask = gG[k].abs() > 1e-4 * gG[k].abs().max()
            if mask.any():
                d_ref = (Gn[k] - Gsd[k])[mask]; d_gpu = (p.detach().cpu() - Gsd[k])[mask]
                agree = torch.isclose(d_gpu, d_ref, rtol=0.05, atol=2e-7).float().mean().item()
                assert agree >= 0.98, f'{k}: only {agree:.3f} of the Adam updates agree'
    print('worst grad rel err', worst)


def test_static_shapes_and_graph_replay_match_eager(dev):
    """module.static_shapes (sync-free heads / masked losses) gives the reference-shaped path's gradients on a ragged mask,
    and a hipGraph replay of each phase reproduces the eager gradients bit-for-bit-ish (dropout off)."""
    from layoutdetr_amd.training import training_loop as tl
    from layoutdetr_amd.training.loss import StyleGAN2Loss
    from layoutdetr_amd.training.networks_detr import TextFeatures
    G, D = _make(dev, seed=5)
    bt, zg, zd = _batch(2, 64, seed=6)
    bt['padding_mask'][1, 3:] = True
    G.eval().requires_grad_(False).to(dev); D.eval().requires_grad_(False).to(dev)
    pG = tl.Phase('Gmain', G, lr=0.0); pD = tl.Phase('Dmain', D, lr=0.0)
    loss = StyleGAN2Loss(dev, G, D)
    dp = tl.DataParallelStep(world_size=1)
    batch = dict(bbox_real=bt['bbox_real'].to(dev), bbox_class=bt['bbox_class'].to(dev),
                 bbox_text=TextFeatures(bt['text_feat'].to(dev), bt['text_len'].to(dev)), bbox_patch=torch.zeros(2, 9, 1, 1, 1, device=dev),
                 padding_mask=bt['padding_mask'].to(dev), background=bt['background'].to(dev), real_c=torch.zeros(2, 0, device=dev),
                 gen_c=torch.zeros(2, 0, device=dev))
    grads = {}
    orig = dp.apply

    def spy(phase, **kw):
        grads.setdefault(phase.name, []).append(phase.fm.gflat.detach().clone())
        orig(phase, **kw)
    dp.apply = spy
    z = [zg.to(dev), zd.to(dev)]
    tl.training_iteration(loss, [pG, pD], dp, batch, 2, z)                 # reference-shaped (gather) path
    G.static_shapes = D.static_shapes = True
    tl.training_iteration(loss, [pG, pD], dp, batch, 2, z)                 # static path, eager
    for name in ('Gmain', 'Dmain'):
        a, b = grads[name]
        e = ((a - b).norm() / b.norm()).item()   # L2-relative: robust to the odd flipped ReLU mask
        # run-to-run noise of the eager path itself is ~5e-4 here (fp32 atomics reorder sums by 1e-7, which flips the odd ReLU mask)
        assert e <= 5e-3, f'{name}: static vs gather rel err {e:.3e}'
    # graph replay (gen_z is drawn inside the graph, so compare two replays with the generator state restored)
    gi = tl.GraphedIteration(loss, [pG, pD], dp, batch, 2, 4)
    st = torch.cuda.get_rng_state(dev)
    gi.run(); torch.cuda.synchronize()
    torch.cuda.set_rng_state(st, dev)
    gi.run(); torch.cuda.synchronize()
    for name in ('Gmain', 'Dmain'):
        a, b = grads[name][-2], grads[name][-1]
        assert torch.isfinite(a).all() and a.abs().sum() > 0
        assert ((a - b).norm() / b.norm()).item() < 5e-3, name


# ------------------------------------------------------------------------------------------ BERT text encoder
@pytest.mark.gpu
@pytest.mark.parametrize('tag', ['', '_dh64'])
def test_bert_text_encoder_vs_reference_golden(dev, tag):
    """HIP text encoder (packed qkv GEMM, fused attention, GELU epilogue, fused add+LN) against the reference modules' output."""
    from layoutdetr_amd.training import med
    d = np.load(os.path.join(G_DIR, f"bert_text{tag}.npz"))
    sd = {k[3:]: torch.from_numpy(d[k]) for k in d.files if k.startswith('sd/')}
    hid = sd['embeddings.word_embeddings.weight'].shape[1]
    cfg = med.BertConfig(vocab_size=sd['embeddings.word_embeddings.weight'].shape[0], hidden_size=hid, num_hidden_layers=2,
                         num_attention_heads=int(d['num_heads']), intermediate_size=sd['encoder.layer.0.intermediate.dense.weight'].shape[0],
                         max_position_embeddings=sd['embeddings.position_embeddings.weight'].shape[0], add_cross_attention=False)
    m = med.BertModel(cfg).eval().requires_grad_(False)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and not [k for k in missing if 'crossattention' not in k]
    m.to(dev)
    out = m(torch.from_numpy(d['input_ids']).to(dev), attention_mask=torch.from_numpy(d['attention_mask']).to(dev), return_dict=True, mode='text')
    ref = torch.from_numpy(d['last_hidden_state'])
    err = (out.last_hidden_state.cpu() - ref).abs().max().item() / ref.abs().max().item()
    assert err <= 1e-5, err
    with pytest.raises(RuntimeError):
        med.BertModel(cfg).to(dev)(torch.from_numpy(d['input_ids']).to(dev))     # parameters still require grad: forward-only module


@pytest.mark.gpu
def test_bert_text_encoder_hot_path_shape_vs_oracle(dev):
    """The hot path's configuration class (hidden 768, 4 heads x 192, GELU 3072) at 2 layers, 40 tokens, ragged masks."""
    from layoutdetr_amd.training import med
    from oracle import bert_ref
    torch.manual_seed(91)
    cfg = med.BertConfig(vocab_size=300, hidden_size=768, num_hidden_layers=2, num_attention_heads=4, intermediate_size=3072,
                         max_position_embeddings=64, add_cross_attention=True)
    m = med.BertModel(cfg).eval().requires_grad_(False)
    for n, p in m.named_parameters():
        p.data.normal_(0, 0.03)
        if 'LayerNorm.weight' in n:
            p.data.add_(1.0)
    B, T = 6, 40
    ids = torch.randint(1, 300, (B, T)); am = torch.ones(B, T, dtype=torch.long)
    am[0, 17:] = 0; am[3, 1:] = 0; am[5, 33:] = 0
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    ref = bert_ref.bert_text_forward(sd, 4, ids, am)
    m.to(dev)
    out = m(ids.to(dev), attention_mask=am.to(dev)).last_hidden_state.cpu()
    keep = am.bool()
    err = (out - ref)[keep].abs().max().item() / ref.abs().max().item()
    assert err <= 2e-5, err
    # train mode draws dropout masks (attention probabilities + hidden states): finite, different from eval, same CLS scale
    m.train()
    out_t = m(ids.to(dev), attention_mask=am.to(dev)).last_hidden_state.cpu()
    assert torch.isfinite(out_t).all() and (out_t - out).abs().max().item() > 1e-3


@pytest.mark.gpu
def test_generator_text_mode_encoder_equals_features_mode(dev):
    """G with the in-module text encoder (TextTokens in) == G fed the same encoder's CLS features (TextFeatures in),
    and a full G+D iteration runs with tokens (text encoder frozen inside the flat-parameter step)."""
    from layoutdetr_amd.training.networks_detr import Discriminator, Generator, TextFeatures, TextTokens
    from layoutdetr_amd.training import training_loop as tl
    from layoutdetr_amd.training.loss import StyleGAN2Loss
    torch.manual_seed(12)
    bg, B, N, T = 64, 2, 9, 24
    kw = dict(num_bbox_labels=8, img_channels=3, img_height=bg, img_width=bg, c_dim=0, background_size=bg, bert_f_dim=768, im_f_dim=512,
              bert_num_encoder_layers=2, bert_num_heads=4)
    G = Generator(z_dim=4, text_mode='encoder', **kw).eval().requires_grad_(False).to(dev)
    D = Discriminator(text_mode='encoder', **kw).eval().requires_grad_(False).to(dev)
    ids = torch.randint(1, 30000, (B, N, T), device=dev); am = torch.ones(B, N, T, dtype=torch.long, device=dev)
    am[0, 2, 9:] = 0; am[1, :, 15:] = 0
    text_len = torch.randint(1, 40, (B, N), device=dev)
    toks = TextTokens(ids, am, text_len)
    xy = torch.rand(B, N, 2, device=dev) * 0.6 + 0.2; wh = torch.rand(B, N, 2, device=dev) * 0.35 + 0.05
    bbox = torch.cat([xy, wh], -1); cls = torch.randint(0, 8, (B, N), device=dev)
    patch = torch.zeros(B, N, 1, 1, 1, device=dev).expand(B, N, 3, 8, 8)
    pm = torch.zeros(B, N, dtype=torch.bool, device=dev); back = torch.randn(B, 3, bg, bg, device=dev)
    z = torch.randn(B, N, 4, device=dev)
    with torch.no_grad():
        out_tok = G(z, cls, bbox, toks, patch, pm, back, None)
        feat = G.text_encoder(ids.reshape(B * N, T), attention_mask=am.reshape(B * N, T)).last_hidden_state[:, 0].reshape(B, N, -1)
        out_feat = G(z, cls, bbox, TextFeatures(feat, text_len), patch, pm, back, None)
    assert torch.equal(out_tok, out_feat)
    G.train(); D.train(); G.static_shapes = D.static_shapes = True
    pG = tl.Phase('Gmain', G, lr=1e-5); pD = tl.Phase('Dmain', D, lr=1e-5)
    batch = dict(bbox_real=bbox, bbox_class=cls, bbox_text=toks, bbox_patch=patch, padding_mask=pm, background=back,
                 real_c=torch.zeros(B, 0, device=dev), gen_c=torch.zeros(B, 0, device=dev))
    w0 = G.text_encoder.encoder.layer[0].intermediate.dense.weight.clone()
    tl.training_iteration(StyleGAN2Loss(dev, G, D), [pG, pD], tl.DataParallelStep(1), batch, B, [z, z])
    torch.cuda.synchronize()
    assert torch.equal(G.text_encoder.encoder.layer[0].intermediate.dense.weight, w0), 'frozen text encoder moved'
    assert all(torch.isfinite(p).all() for p in G.parameters())


@pytest.mark.gpu
def test_iteration_level_D_trunk_sharing_matches_reference_call_pattern(dev):
    """share_D_trunk='iteration' (one D-trunk evaluation per iteration, its backward in Dmain) against the reference's call
    pattern (share_D_trunk=False: the trunk evaluated in Gmain and twice in Dmain): same gradients for both phases, eager and
    as replayed hipGraphs (trunk graph + phase graphs in one memory pool)."""
    from layoutdetr_amd.training import training_loop as tl
    from layoutdetr_amd.training.loss import StyleGAN2Loss
    from layoutdetr_amd.training.networks_detr import TextFeatures
    G, D = _make(dev, seed=7)
    bt, zg, zd = _batch(2, 64, seed=8)
    G.eval().requires_grad_(False).to(dev); D.eval().requires_grad_(False).to(dev)
    G.static_shapes = D.static_shapes = True
    pG = tl.Phase('Gmain', G, lr=0.0); pD = tl.Phase('Dmain', D, lr=0.0)
    dp = tl.DataParallelStep(world_size=1)
    batch = dict(bbox_real=bt['bbox_real'].to(dev), bbox_class=bt['bbox_class'].to(dev),
                 bbox_text=TextFeatures(bt['text_feat'].to(dev), bt['text_len'].to(dev)), bbox_patch=torch.zeros(2, 9, 1, 1, 1, device=dev),
                 padding_mask=bt['padding_mask'].to(dev), background=bt['background'].to(dev), real_c=torch.zeros(2, 0, device=dev),
                 gen_c=torch.zeros(2, 0, device=dev))
    grads = {}
    orig = dp.apply

    def spy(phase, **kw):
        grads.setdefault(phase.name, []).append(phase.fm.gflat.detach().clone())
        orig(phase, **kw)
    dp.apply = spy
    z = [zg.to(dev), zd.to(dev)]
    tl.training_iteration(StyleGAN2Loss(dev, G, D, share_D_trunk=False), [pG, pD], dp, batch, 2, z)
    loss_it = StyleGAN2Loss(dev, G, D, share_D_trunk='iteration')
    tl.training_iteration(loss_it, [pG, pD], dp, batch, 2, z)
    assert not loss_it._trunk_cache, 'trunk cache must be consumed by Dmain'
    for name in ('Gmain', 'Dmain'):
        a, b = grads[name]
        e = ((a - b).norm() / a.norm()).item()
        assert e <= 5e-3, f'{name}: iteration-level sharing vs reference call pattern: rel err {e:.3e}'
    # graphed: gen_z is drawn inside the graphs -> compare two replays under the same generator state
    gi = tl.GraphedIteration(loss_it, [pG, pD], dp, batch, 2, 4)
    assert gi.pre_graph is not None
    st = torch.cuda.get_rng_state(dev)
    gi.run(); torch.cuda.synchronize()
    n0 = len(grads['Dmain'])
    torch.cuda.set_rng_state(st, dev)
    gi.run(); torch.cuda.synchronize()
    for name in ('Gmain', 'Dmain'):
        a, b = grads[name][-2], grads[name][-1]
        e = ((a - b).norm() / a.norm()).item()
        assert e <= 5e-3 and torch.isfinite(a).all(), f'{name}: graph replays differ: {e:.3e}'
    assert len(grads['Dmain']) == n0 + 1


@pytest.mark.gpu
def test_bert_lm_decoder_vs_reference_golden(dev):
    """HIP LM decoder (causal attention fwd/bwd, GELU fwd/grad, tied LM head, label-smoothed CE) against the reference pieces:
    loss, shifted logits and the gradient of every parameter."""
    from layoutdetr_amd.training import med
    d = np.load(os.path.join(G_DIR, 'bert_lm.npz'))
    sd = {k[3:]: torch.from_numpy(d[k]) for k in d.files if k.startswith('sd/')}
    V, hid = sd['bert.embeddings.word_embeddings.weight'].shape
    cfg = med.BertConfig(vocab_size=V, hidden_size=hid, num_hidden_layers=2, num_attention_heads=int(d['num_heads']),
                         intermediate_size=sd['bert.encoder.layer.0.intermediate.dense.weight'].shape[0],
                         max_position_embeddings=sd['bert.embeddings.position_embeddings.weight'].shape[0], add_cross_attention=False)
    m = med.BertLMHeadModel(cfg).eval()
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and set(missing) <= {'bert.embeddings.position_ids', 'cls.predictions.decoder.weight'}, (missing, unexpected)
    m.to(dev)
    out = m(torch.from_numpy(d['input_ids']).to(dev), attention_mask=torch.from_numpy(d['attention_mask']).to(dev),
            encoder_hidden_states=torch.zeros(4, 1, hid, device=dev), labels=torch.from_numpy(d['labels']).to(dev), return_dict=True, mode='text')
    out.loss.backward()
    assert abs(out.loss.item() - float(d['loss'])) <= 2e-5, (out.loss.item(), float(d['loss']))
    ref_logits = torch.from_numpy(d['logits'])
    assert (out.logits.detach().cpu() - ref_logits).abs().max().item() <= 2e-5 * ref_logits.abs().max().item()
    gmax = max(float(np.abs(d['grad/' + k]).max()) for k in sd)
    named = dict(m.named_parameters())
    for k in sd:
        g = named[k].grad
        assert g is not None, k
        err = (g.cpu() - torch.from_numpy(d['grad/' + k])).abs().max().item()
        assert err <= 3e-5 * gmax, f'{k}: {err / gmax:.3e}'


@pytest.mark.gpu
def test_text_mode_encoder_lm_full_iteration(dev):
    """text_mode='encoder+lm': G's reconstructor returns the LM decoder's loss (= oracle on the same tokens / masks / padded
    slots, static and gather formulations alike) and a full G+D iteration trains the decoder while the encoder stays frozen."""
    from layoutdetr_amd.training.networks_detr import Discriminator, Generator, TextTokens
    from layoutdetr_amd.training import training_loop as tl
    from layoutdetr_amd.training.loss import StyleGAN2Loss
    from oracle import bert_ref
    torch.manual_seed(14)
    bg, B, N, T = 64, 2, 9, 16
    kw = dict(num_bbox_labels=8, img_channels=3, img_height=bg, img_width=bg, c_dim=0, background_size=bg, bert_f_dim=768, im_f_dim=512,
              bert_num_encoder_layers=1, bert_num_decoder_layers=2, bert_num_heads=4, text_mode='encoder+lm')
    G = Generator(z_dim=4, **kw).eval().requires_grad_(False).to(dev)
    D = Discriminator(**kw).eval().requires_grad_(False).to(dev)
    ids = torch.randint(1, 30000, (B, N, T), device=dev); am = torch.ones(B, N, T, dtype=torch.long, device=dev)
    am[0, 2, 9:] = 0; am[1, :, 11:] = 0; ids[am == 0] = 0
    toks = TextTokens(ids, am, torch.randint(1, 40, (B, N), device=dev))
    xy = torch.rand(B, N, 2, device=dev) * 0.6 + 0.2; wh = torch.rand(B, N, 2, device=dev) * 0.35 + 0.05
    bbox = torch.cat([xy, wh], -1); cls = torch.randint(0, 8, (B, N), device=dev)
    patch = torch.zeros(B, N, 1, 1, 1, device=dev).expand(B, N, 3, 8, 8)
    pm = torch.zeros(B, N, dtype=torch.bool, device=dev); pm[1, 6:] = True
    back = torch.randn(B, 3, bg, bg, device=dev); z = torch.randn(B, N, 4, device=dev)
    # oracle value of loss_lm
    sd = {k: v.detach().cpu() for k, v in G.text_decoder.state_dict().items()}
    dec_ids = ids.reshape(B * N, T).cpu().clone(); dec_ids[:, 0] = toks.bos_token_id
    tg = dec_ids.masked_fill(dec_ids == 0, -100); keep = ~pm.reshape(-1).cpu()
    ref, _ = bert_ref.bert_lm_loss(sd, 4, dec_ids[keep], am.reshape(B * N, T).cpu()[keep], tg[keep])
    with torch.no_grad():
        lm_gather = G(z, cls, bbox, toks, patch, pm, back, None, reconst=True)[3]
        G.static_shapes = True
        lm_static = G(z, cls, bbox, toks, patch, pm, back, None, reconst=True)[3]
    assert abs(lm_gather.item() - ref.item()) <= 2e-4 * abs(ref.item()), (lm_gather.item(), ref.item())
    assert abs(lm_static.item() - ref.item()) <= 2e-4 * abs(ref.item()), (lm_static.item(), ref.item())
    # one training iteration: decoder moves, encoder does not
    G.train(); D.train(); D.static_shapes = True
    pG = tl.Phase('Gmain', G, lr=1e-4); pD = tl.Phase('Dmain', D, lr=1e-4)
    batch = dict(bbox_real=bbox, bbox_class=cls, bbox_text=toks, bbox_patch=patch, padding_mask=pm, background=back,
                 real_c=torch.zeros(B, 0, device=dev), gen_c=torch.zeros(B, 0, device=dev))
    enc0 = G.text_encoder.encoder.layer[0].intermediate.dense.weight.clone()
    dec0 = {n: p.detach().clone() for n, p in G.text_decoder.named_parameters() if 'crossattention' not in n}
    ddec0 = D.text_decoder.cls.predictions.transform.dense.weight.detach().clone()
    loss = StyleGAN2Loss(dev, G, D)
    tl.training_iteration(loss, [pG, pD], tl.DataParallelStep(1), batch, B, [z, z])
    torch.cuda.synchronize()
    assert torch.equal(G.text_encoder.encoder.layer[0].intermediate.dense.weight, enc0)
    moved = [n for n, p in G.text_decoder.named_parameters() if n in dec0 and not torch.equal(p.detach(), dec0[n])]
    assert len(moved) >= len(dec0) - 4, f'only {len(moved)} of {len(dec0)} decoder tensors were updated'   # key biases have zero gradient
    assert not torch.equal(D.text_decoder.cls.predictions.transform.dense.weight.detach(), ddec0)
    assert loss.last['loss_Ggen_text_rec'].abs().item() > 0


@pytest.mark.gpu
@pytest.mark.parametrize('workspace', [True, False])
def test_graph_replay_with_lm_decoder_survives_allocator_churn(dev, workspace):
    """Guards the two hipGraph-replay faults found at the B=16 hot-path size (this
    smaller case did not trip the old code every time): (1) memset nodes (the zero-fill of the atomic split-K path) and (2) aten's
    sort-based embedding backward (> 3072 tokens) replayed with garbage.  Both are own kernels now.  Replays must stay finite and
    reproducible when eager code frees, unmaps and overwrites allocator blocks in between, with the split-K fix-up scratch
    (default) and on the fp32-atomic path."""
    from layoutdetr_amd.hip import core
    from layoutdetr_amd.training.networks_detr import Discriminator, Generator, TextTokens
    from layoutdetr_amd.training import training_loop as tl
    from layoutdetr_amd.training.loss import StyleGAN2Loss
    torch.manual_seed(21)
    bg, B, N, T = 64, 4, 9, 96       # 3456 tokens
    kw = dict(num_bbox_labels=8, img_channels=3, img_height=bg, img_width=bg, c_dim=0, background_size=bg, bert_f_dim=768, im_f_dim=512,
              bert_num_encoder_layers=1, bert_num_decoder_layers=1, bert_num_heads=4, text_mode='encoder+lm')
    try:
        if not workspace:
            core.disable_splitk_workspace()
        G = Generator(z_dim=4, **kw).eval().requires_grad_(False).to(dev)
        D = Discriminator(**kw).eval().requires_grad_(False).to(dev)
        G.static_shapes = D.static_shapes = True
        ids = torch.randint(1, 30000, (B, N, T), device=dev); am = torch.ones(B, N, T, dtype=torch.long, device=dev)
        am[:, :, 80:] = 0; ids[am == 0] = 0
        toks = TextTokens(ids, am, torch.randint(1, 40, (B, N), device=dev))
        xy = torch.rand(B, N, 2, device=dev) * 0.6 + 0.2; wh = torch.rand(B, N, 2, device=dev) * 0.35 + 0.05
        pm = torch.zeros(B, N, dtype=torch.bool, device=dev); pm[1, 6:] = True
        batch = dict(bbox_real=torch.cat([xy, wh], -1), bbox_class=torch.randint(0, 8, (B, N), device=dev), bbox_text=toks,
                     bbox_patch=torch.zeros(B, N, 3, 8, 8, device=dev), padding_mask=pm, background=torch.randn(B, 3, bg, bg, device=dev),
                     real_c=torch.zeros(B, 0, device=dev), gen_c=torch.zeros(B, 0, device=dev))
        pG = tl.Phase('Gmain', G, lr=0.0); pD = tl.Phase('Dmain', D, lr=0.0)
        loss = StyleGAN2Loss(dev, G, D); dp = tl.DataParallelStep(1)
        grads = {}
        orig = dp.apply

        def spy(phase, **kw):
            grads.setdefault(phase.name, []).append(phase.fm.gflat.detach().clone())
            orig(phase, **kw)
        dp.apply = spy
        side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
        st = torch.cuda.get_rng_state(dev)
        with torch.cuda.stream(side):
            for _ in range(2):
                torch.cuda.set_rng_state(st, dev)
                tl.training_iteration(loss, [pG, pD], dp, batch, B, [torch.randn(B, N, 4, device=dev) for _ in range(2)])
        torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
        gi = tl.GraphedIteration(loss, [pG, pD], dp, batch, B, 4, capture_stream=side)
        torch.cuda.synchronize()
        for _ in range(3):
            import gc
            gc.collect(); torch.cuda.empty_cache()
            junk = [torch.full((n,), float('nan'), device=dev) for n in (1 << 24, 1 << 20, 1 << 16, 3000, 512)]   # scribble over freed blocks
            del junk
            torch.cuda.set_rng_state(st, dev)
            gi.run(); torch.cuda.synchronize()
        for name in ('Gmain', 'Dmain'):
            eager, replays = grads[name][1], grads[name][2:]
            assert len(replays) == 3
            for r in replays:
                assert torch.isfinite(r).all(), name
                # gen_z differs between the eager run and the replays (drawn inside the graph), so compare replays with each other
                # tightly and with the eager run by magnitude
                assert ((r - replays[0]).norm() / replays[0].norm()).item() < 5e-3, name
                assert 0.2 < (r.norm() / eager.norm()).item() < 5.0, name
    finally:
        core.enable_splitk_workspace()


# ------------------------------------------------------------------------------------------ staged backward / overlapped exchange
def _iteration_grads(dev, mode, seed=5, share=True, stages=3):
    """Gradients of both phases of one iteration at B=2, 64x64 (eval: dropout off).  mode: 'plain' | 'staged' | 'graph-staged';
    stages: 3 (layer1-2 | layer3-4 | rest), 2 (trunk | rest) or None = training_loop.backward_stage_count's rule for this batch."""
    prev = os.environ.get('LDETR_BACKWARD_STAGES')
    if stages is None:
        os.environ.pop('LDETR_BACKWARD_STAGES', None)
    else:
        os.environ['LDETR_BACKWARD_STAGES'] = str(stages)
    try:
        return _iteration_grads_impl(dev, mode, seed, share)
    finally:
        if prev is None:
            os.environ.pop('LDETR_BACKWARD_STAGES', None)
        else:
            os.environ['LDETR_BACKWARD_STAGES'] = prev


def _iteration_grads_impl(dev, mode, seed, share):
    from layoutdetr_amd.training import training_loop as tl
    from layoutdetr_amd.training.loss import StyleGAN2Loss
    from layoutdetr_amd.training.networks_detr import TextFeatures
    G, D = _make(dev, seed=seed)
    bt, zg, zd = _batch(2, 64, seed=seed + 1)
    bt['padding_mask'][:] = False
    G.eval().requires_grad_(False).to(dev); D.eval().requires_grad_(False).to(dev)
    G.static_shapes = D.static_shapes = True
    pG = tl.Phase('Gmain', G, lr=0.0); pD = tl.Phase('Dmain', D, lr=0.0)
    loss = StyleGAN2Loss(dev, G, D, share_D_trunk=share)
    dp = tl.DataParallelStep(world_size=1)
    batch = dict(bbox_real=bt['bbox_real'].to(dev), bbox_class=bt['bbox_class'].to(dev),
                 bbox_text=TextFeatures(bt['text_feat'].to(dev), bt['text_len'].to(dev)), bbox_patch=torch.zeros(2, 9, 1, 1, 1, device=dev),
                 padding_mask=bt['padding_mask'].to(dev), background=bt['background'].to(dev), real_c=torch.zeros(2, 0, device=dev),
                 gen_c=torch.zeros(2, 0, device=dev))
    grads = {}
    orig = dp.apply

    def spy(phase, exchanged=False, **kw):
        grads[phase.name] = (phase.fm.gflat.detach().clone(), exchanged)
        orig(phase, exchanged=exchanged, **kw)
    dp.apply = spy
    if mode == 'graph-staged':
        side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            tl.training_iteration(loss, [pG, pD], dp, batch, 2, [zg.to(dev), zd.to(dev)], overlap=True)
        torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
        gi = tl.GraphedIteration(loss, [pG, pD], dp, batch, 2, 4, capture_stream=side, overlap=True)
        nst = tl.backward_stage_count(2)
        assert [len(c) for c in gi.graphs] == [nst, nst], f'each phase must be {nst} chained stage graphs'
        # the graph draws its own gen_z: feed the same latent through the generator state instead
        torch.manual_seed(99); gi.run(); torch.cuda.synchronize()
        a = {k: v[0].clone() for k, v in grads.items()}
        torch.manual_seed(99); gi.run(); torch.cuda.synchronize()
        for k in a:      # replays are reproducible (same latent draw) and finite
            assert torch.isfinite(grads[k][0]).all() and (grads[k][0] - a[k]).abs().max() <= 1e-3 * a[k].abs().max()   # fp32 atomics: run-to-run ~1e-4
        return grads, pG.fm
    tl.training_iteration(loss, [pG, pD], dp, batch, 2, [zg.to(dev), zd.to(dev)], overlap=(mode == 'staged'))
    torch.cuda.synchronize()
    return grads, pG.fm


def test_staged_backward_equals_plain_backward(dev):
    """The three-stage backward that feeds the overlapped gradient exchange (trunk cut at layer2|layer3 and trunk|rest) produces the
    gradients of the single backward pass; the flat-buffer segments of the stages tile the buffer exactly."""
    plain, fm = _iteration_grads(dev, 'plain')
    staged, _ = _iteration_grads(dev, 'staged')
    segs = fm.stage_segments()
    ranges = sorted(r for st in segs for r in st)
    assert ranges[0][0] == 0 and ranges[-1][1] == fm.total and all(a[1] == b[0] for a, b in zip(ranges[:-1], ranges[1:])), 'segments must tile the buffer'
    (t_lo, _), = segs[2]; (_, t_hi), = segs[1]
    names = [n for n in fm.names if n.startswith('backbone.0.body.')]
    assert all(t_lo <= fm.offsets[fm.names.index(n)] < t_hi for n in names) and len(names) == 53
    for k in ('Gmain', 'Dmain'):
        assert plain[k][1] is False and staged[k][1] is True
        check(staged[k][0], plain[k][0], 2e-4, f'{k} flat gradient, staged vs plain')   # fp32 atomics in the split-K weight gradients: run-to-run ~5e-5
    _iteration_grads(dev, 'graph-staged')
    # iteration-level D-trunk sharing (D's trunk evaluated once, before the phases): its cuts are recorded there and Dmain's staged
    # backward continues from them
    it_staged, _ = _iteration_grads(dev, 'staged', share='iteration')
    for k in ('Gmain', 'Dmain'):
        assert it_staged[k][1] is True
        check(it_staged[k][0], plain[k][0], 2e-4, f'{k} flat gradient, iteration-shared + staged vs plain')
    _iteration_grads(dev, 'graph-staged', share='iteration')
    # two stages (trunk | rest): what backward_stage_count picks at <= 4 samples per GPU, where three stage graphs are too short to hide the
    # host's issue latency between replays
    from layoutdetr_amd.training import training_loop as tl
    assert tl.backward_stage_count(2) == 2 and tl.backward_stage_count(4) == 2 and tl.backward_stage_count(16) == 3
    segs2 = fm.stage_segments(2)
    assert len(segs2) == 2 and segs2[0] == segs[0] and segs2[1] == [(t_lo, t_hi)]
    for share in (True, 'iteration'):
        two, _ = _iteration_grads(dev, 'staged', share=share, stages=None)
        for k in ('Gmain', 'Dmain'):
            assert two[k][1] is True
            check(two[k][0], plain[k][0], 2e-4, f'{k} flat gradient, two-stage (share={share}) vs plain')
        _iteration_grads(dev, 'graph-staged', share=share, stages=None)


def test_two_rank_sharded_step_equals_one_rank_global_batch(dev):
    """2 ranks x 2 samples (all 9 slots valid) == 1 rank x 4 samples after one iteration: staged backward, segment-wise overlapped
    exchange on the communication stream, /world + nan_to_num fused into Adam (fuse_sanitize=1, gscale=1/2).  With >= 2 GPUs the
    ranks sit on two devices and exchange through RCCL; on a 1-GPU box both ranks share cuda:0 and exchange through gloo (same
    rank logic, same streams)."""
    import subprocess
    import sys
    code = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["LDETR_ROOT"]); sys.path.insert(0, os.path.join(os.environ["LDETR_ROOT"], "tests"))
import test_model_gpu as T
from layoutdetr_amd.training import training_loop as tl
from layoutdetr_amd.training.loss import StyleGAN2Loss
from layoutdetr_amd.training.networks_detr import TextFeatures
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
two = torch.cuda.device_count() >= 2
torch.cuda.set_device(rank if two else 0); dev = torch.device("cuda", rank if two else 0)
if world > 1:
    dist.init_process_group("nccl", device_id=dev) if two else dist.init_process_group("gloo")
G, D = T._make(dev, seed=5)
bt, zg, zd = T._batch(4, 64, seed=6); bt["padding_mask"][:] = False
sl = slice(rank * (4 // world), (rank + 1) * (4 // world))
G.eval().requires_grad_(False).to(dev); D.eval().requires_grad_(False).to(dev)
pG = tl.Phase("Gmain", G, lr=1e-3); pD = tl.Phase("Dmain", D, lr=1e-3)
dp = tl.DataParallelStep(world_size=world)
n = 4 // world
batch = dict(bbox_real=bt["bbox_real"][sl].to(dev), bbox_class=bt["bbox_class"][sl].to(dev), bbox_text=TextFeatures(bt["text_feat"][sl].to(dev), bt["text_len"][sl].to(dev)),
             bbox_patch=torch.zeros(n, 9, 1, 1, 1, device=dev), padding_mask=bt["padding_mask"][sl].to(dev), background=bt["background"][sl].to(dev),
             real_c=torch.zeros(n, 0, device=dev), gen_c=torch.zeros(n, 0, device=dev))
tl.training_iteration(StyleGAN2Loss(dev, G, D), [pG, pD], dp, batch, n, [zg[sl].to(dev), zd[sl].to(dev)])
torch.cuda.synchronize()
if rank == 0:
    torch.save(dict(G=pG.fm.flat.cpu(), D=pD.fm.flat.cpu()), os.environ["LDETR_OUT"])
if world > 1:
    dist.barrier(); dist.destroy_process_group()
'''
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for world in (1, 2):
        out = tempfile.mktemp(suffix='.pt')
        env = dict(os.environ, LDETR_ROOT=root, LDETR_OUT=out, HSA_ENABLE_IPC_MODE_LEGACY='0')
        if world == 1:
            env.update(RANK='0', WORLD_SIZE='1')
            subprocess.check_call([sys.executable, '-c', code], env=env)
        else:
            subprocess.check_call([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1',
                                   '--master-port', '29533', '--no-python', sys.executable, '-c', code], env=env)
        outs.append(torch.load(out))
    for k in ('G', 'D'):
        # the mean over 4 samples == the mean of two 2-sample means (equal valid-slot counts per rank: SURVEY 8e); Adam's first step moves
        # every parameter by ~lr*sign(g), so compare where |g| is not rounding noise: >= 99 % of the updates agree to 5 %
        a, b = outs[0][k], outs[1][k]
        agree = torch.isclose(a, b, rtol=1e-4, atol=2e-5).float().mean().item()
        assert agree >= 0.99, f'{k}: only {agree:.4f} of the parameters agree between 1 rank x 4 and 2 ranks x 2'


def test_two_ranks_overlapped_exchange_equals_exchange_after_backward(dev):
    """2 ranks x 2 samples, three iterations each: the staged backward with the segment-wise exchange launched behind every stage on the communication
    stream (training_loop.staged_backward + DataParallelStep.exchange_async: what GraphedIteration(overlap=True) replays) hands Adam the SAME
    gradients as one all-reduce after the whole backward (the reference's order, training_loop.py:303-312; bench.py --no-overlap): the exchanged
    flat gradient of the last iteration and Adam's second moment (which has seen all three) are compared, at lr = 0 so that the three iterations see
    the same weights in both runs (with a real step the two runs drift apart chaotically from the order of the weight-gradient atomics alone).
    RCCL with >= 2 GPUs, gloo with both ranks on one device otherwise."""
    import subprocess
    import sys
    import tempfile
    code = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["LDETR_ROOT"]); sys.path.insert(0, os.path.join(os.environ["LDETR_ROOT"], "tests"))
import test_model_gpu as T
from layoutdetr_amd.training import training_loop as tl
from layoutdetr_amd.training.loss import StyleGAN2Loss
from layoutdetr_amd.training.networks_detr import TextFeatures
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
two = torch.cuda.device_count() >= 2
torch.cuda.set_device(rank if two else 0); dev = torch.device("cuda", rank if two else 0)
dist.init_process_group("nccl", device_id=dev) if two else dist.init_process_group("gloo")
G, D = T._make(dev, seed=5)
bt, zg, zd = T._batch(4, 64, seed=6); bt["padding_mask"][:] = False
n = 4 // world
sl = slice(rank * n, (rank + 1) * n)
G.eval().requires_grad_(False).to(dev); D.eval().requires_grad_(False).to(dev)
pG = tl.Phase("Gmain", G, lr=0.0); pD = tl.Phase("Dmain", D, lr=0.0)
dp = tl.DataParallelStep(world_size=world)
batch = dict(bbox_real=bt["bbox_real"][sl].to(dev), bbox_class=bt["bbox_class"][sl].to(dev), bbox_text=TextFeatures(bt["text_feat"][sl].to(dev), bt["text_len"][sl].to(dev)),
             bbox_patch=torch.zeros(n, 9, 1, 1, 1, device=dev), padding_mask=bt["padding_mask"][sl].to(dev), background=bt["background"][sl].to(dev),
             real_c=torch.zeros(n, 0, device=dev), gen_c=torch.zeros(n, 0, device=dev))
loss = StyleGAN2Loss(dev, G, D, share_D_trunk="iteration")
for it in range(3):
    tl.training_iteration(loss, [pG, pD], dp, batch, n, [zg[sl].to(dev), zd[sl].to(dev)], overlap=os.environ["LDETR_OVERLAP"] == "1")
torch.cuda.synchronize()
if rank == 0:
    torch.save(dict(G=pG.fm.gflat.cpu(), D=pD.fm.gflat.cpu(), vG=pG.v.cpu(), vD=pD.v.cpu()), os.environ["LDETR_OUT"])
dist.barrier(); dist.destroy_process_group()
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for overlap in ('1', '0'):
        out = tempfile.mktemp(suffix='.pt')
        env = dict(os.environ, LDETR_ROOT=root, LDETR_OUT=out, LDETR_OVERLAP=overlap, HSA_ENABLE_IPC_MODE_LEGACY='0')
        subprocess.check_call([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1',
                               '--master-port', '29541', '--no-python', sys.executable, '-c', code], env=env)
        outs.append(torch.load(out))
    for k in ('G', 'D', 'vG', 'vD'):
        # what differs between two separate runs is only the order in which the weight-gradient atomics landed
        a, b = outs[0][k], outs[1][k]
        assert a.abs().max() > 0
        agree = torch.isclose(a, b, rtol=2e-3, atol=1e-5 * a.abs().max().item()).float().mean().item()
        assert agree >= 0.999, f'{k}: only {agree:.4f} of the entries agree between the overlapped and the after-backward exchange'


def test_decoder_mapping_latent_forms_and_truncation(dev):
    """The Decoder hands its mapping output to the synthesis network as ONE [B, w_dim] latent (no num_ws copies); the reference's
    [B, num_ws, w_dim] form, truncation towards w_avg and the per-layer cutoff (networks_stylegan2.py:951-964) give the same images."""
    from layoutdetr_amd.training.networks_stylegan2 import Decoder
    torch.manual_seed(8)
    m = Decoder(z_dim=32, w_dim=32, img_resolution=16, img_channels=3, use_noise=False, channel_base=256, channel_max=32, num_fp16_res=0,
                conv_clamp=None, fused_modconv_default=False).to(dev).eval()
    m.mapping.w_avg.copy_(torch.randn(32, device=dev) * 0.3)
    z = torch.randn(3, 32, device=dev)
    with torch.no_grad():
        w = m.mapping(z, broadcast=False); ws = m.mapping(z)
        assert w.shape == (3, 32) and ws.shape == (3, m.num_ws, 32) and torch.equal(ws[:, 2], w)
        img = m(z)
        check(m.synthesis(ws), img, 1e-6, 'broadcast latent form')
        wt = m.mapping.w_avg + 0.7 * (w - m.mapping.w_avg)
        check(m(z, truncation_psi=0.7), m.synthesis(wt), 1e-5, 'truncation')
        cut = m.mapping(z, truncation_psi=0.7, truncation_cutoff=3)
        check(cut[:, :3], wt.unsqueeze(1).expand(-1, 3, -1), 1e-5, 'cutoff head'); assert torch.equal(cut[:, 3:], ws[:, 3:])
        check(m(z, truncation_psi=0.7, truncation_cutoff=3), m.synthesis(cut), 1e-6, 'cutoff images')
        before = m.mapping.w_avg.clone()
        m(z, update_emas=True)
        check(m.mapping.w_avg, w.mean(0).lerp(before, m.mapping.w_avg_beta), 1e-6, 'w_avg tracking')


@pytest.mark.gpu
@pytest.mark.parametrize('rows,train', [(1024, False), (2048, True), (640, True)])
def test_large_ffn_node_equals_two_linear_nodes(dev, rows, train):
    """hip.ffn.ffn_large (the feed-forward block of the 64-token encoders above the fused launch's token limit as one autograd node: hidden gradient
    out of the first GEMM's epilogue, both weight gradients as one paired launch) against the two hip.linear nodes it replaces
    (detr_transformer.py:210-214): output, input gradient incl. the residual branch through the alias, and all four parameter gradients, with the
    hidden dropout on (same seed order, same element index -> same mask) and off."""
    from layoutdetr_amd.hip import core, ffn as hffn
    from layoutdetr_amd.hip.linear import linear
    from layoutdetr_amd.training.training_loop import FlatModule

    def rel_err(a, b):
        return ((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-12)).item()
    torch.manual_seed(5)
    lin1, lin2 = torch.nn.Linear(256, 2048).to(dev), torch.nn.Linear(2048, 256).to(dev)
    mod = torch.nn.ModuleList([lin1, lin2])
    fm = FlatModule(mod)
    x0 = torch.randn(rows, 256, device=dev)
    dy = torch.randn(rows, 256, device=dev)
    p = 0.1 if train else 0.0

    def run(fused):
        fm.zero_grad()
        x = x0.clone().requires_grad_(True)
        core._seed_counter[0] = 1000        # the same dropout seed for both paths (the per-iteration device word is not re-drawn in between)
        if fused:
            f, xa = hffn.ffn_large(x, lin1, lin2, p)
        else:
            h, xa = linear(x, lin1.weight, lin1.bias, act=core.ACT_RELU, p_drop=p, passthru=True)
            f = linear(h, lin2.weight, lin2.bias)
        y = xa * 0.5 + f            # the residual branch reads x through the alias
        y.backward(dy)
        return y.detach().clone(), x.grad.clone(), fm.gflat.clone()

    ya, dxa, ga = run(False)
    yb, dxb, gb = run(True)
    assert torch.equal(ya, yb), 'forward: the node launches the same two GEMMs'
    assert rel_err(dxb, dxa) < 2e-6, rel_err(dxb, dxa)
    assert rel_err(gb, ga) < 1e-5, rel_err(gb, ga)
    assert float((ya == 0).float().mean()) < 0.01 and float(gb.abs().max()) > 0
