"""GPU parity of every C-ABI kernel against fp32 CPU references (oracle/ops_ref.py for the reference's
native ops, plain torch fp32 for the ATen-replacing contractions).  Tolerances are written per test;
index outputs (max-pool argmax routing, LSAP) are bit-exact."""
import ctypes
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import ops_ref  # noqa: E402

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def rel_err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


def assert_close(a, b, tol, what=''):
    e = rel_err(a, b)
    assert e <= tol, f'{what}: rel err {e:.3e} > {tol:.1e}'


# ------------------------------------------------------------------------------------------ bias_act
@pytest.mark.parametrize('act', ['linear', 'relu', 'lrelu', 'tanh', 'sigmoid', 'elu', 'selu', 'softplus', 'swish'])
@pytest.mark.parametrize('layout', ['nchw', 'nhwc', 'vec2d'])
def test_bias_act(dev, act, layout):
    from layoutdetr_amd.torch_utils.ops import bias_act
    torch.manual_seed(0)
    shape = (3, 8, 5, 7) if layout != 'vec2d' else (6, 12)
    x = torch.randn(shape) * 2
    b = torch.randn(shape[1])
    xr = x.clone().requires_grad_(True); br = b.clone().requires_grad_(True)
    yr = ops_ref.bias_act(xr, br, act=act, clamp=None)
    g = torch.randn_like(yr)
    yr.backward(g)
    xg = x.to(dev)
    if layout == 'nhwc':
        xg = xg.contiguous(memory_format=torch.channels_last)
    xg.requires_grad_(True); bg = b.to(dev).requires_grad_(True)
    y = bias_act.bias_act(xg, bg, act=act)
    y.backward(g.to(dev))
    assert_close(y, yr, 2e-6, 'y'); assert_close(xg.grad, xr.grad, 5e-6, 'dx'); assert_close(bg.grad, br.grad, 2e-5, 'db')
    # clamp + explicit gain/alpha
    y2 = bias_act.bias_act(xg.detach(), bg.detach(), act=act, gain=0.7, clamp=0.9, alpha=0.1)
    y2r = ops_ref.bias_act(x, b, act=act, gain=0.7, clamp=0.9, alpha=0.1)
    assert_close(y2, y2r, 2e-6, 'clamped')


def test_bias_act_second_order(dev):
    from layoutdetr_amd.torch_utils.ops import bias_act
    torch.manual_seed(1)
    x = torch.randn(4, 6, 3, 3); b = torch.randn(6)
    for act in ['tanh', 'sigmoid', 'softplus', 'swish', 'elu', 'selu']:
        xr = x.clone().requires_grad_(True)
        yr = ops_ref.bias_act(xr, b, act=act)
        (gr,) = torch.autograd.grad(yr.square().sum(), xr, create_graph=True)
        gr.square().sum().backward()
        xg = x.to(dev).requires_grad_(True)
        y = bias_act.bias_act(xg, b.to(dev), act=act)
        (gg,) = torch.autograd.grad(y.square().sum(), xg, create_graph=True)
        gg.square().sum().backward()
        assert_close(xg.grad, xr.grad, 1e-4, f'2nd order {act}')


def test_bias_act_errors(dev):
    from layoutdetr_amd.torch_utils.ops import bias_act
    with pytest.raises(RuntimeError):
        bias_act.bias_act(torch.randn(2, 3), torch.randn(3))  # CPU tensor: no fallback
    with pytest.raises((RuntimeError, AssertionError)):
        bias_act.bias_act(torch.randn(2, 3, device=dev), torch.randn(4, device=dev))
    y = bias_act.bias_act(torch.empty(0, 3, device=dev), torch.randn(3, device=dev), act='lrelu')
    assert y.shape == (0, 3)


# ------------------------------------------------------------------------------------------ upfirdn2d
UPFIRDN_CASES = [
    dict(up=1, down=1, padding=[1, 1, 1, 1], gain=4.0),          # FIR after transposed conv
    dict(up=2, down=1, padding=[2, 1, 2, 1], gain=4.0),          # RGB-skip upsample
    dict(up=1, down=2, padding=[1, 1, 1, 1], gain=1.0),          # its backward / downsample
    dict(up=[2, 1], down=[1, 3], padding=[0, 2, -1, 3], gain=0.5),
    dict(up=3, down=2, padding=[-1, 4, 2, 0], gain=1.5, flip_filter=True),
]


@pytest.mark.parametrize('case', UPFIRDN_CASES)
@pytest.mark.parametrize('layout', ['nchw', 'nhwc'])
def test_upfirdn2d(dev, case, layout):
    from layoutdetr_amd.torch_utils.ops import upfirdn2d
    torch.manual_seed(2)
    x = torch.randn(2, 8, 9, 11)
    f = ops_ref.setup_filter([1, 3, 3, 1])
    f = f + 0.01 * torch.arange(16.).reshape(4, 4)  # break symmetry so flips are detected
    xr = x.clone().requires_grad_(True)
    yr = ops_ref.upfirdn2d(xr, f, **case)
    g = torch.randn_like(yr); yr.backward(g)
    xg = x.to(dev)
    if layout == 'nhwc':
        xg = xg.contiguous(memory_format=torch.channels_last)
    xg.requires_grad_(True)
    y = upfirdn2d.upfirdn2d(xg, f.to(dev), **case)
    assert y.shape == yr.shape
    y.backward(g.to(dev))
    assert_close(y, yr, 3e-6, 'y'); assert_close(xg.grad, xr.grad, 3e-6, 'dx')


@pytest.mark.parametrize('pad,flip', [([1, 1, 1, 1], False), ([2, 2, 2, 2], True), ([3, 0, -1, 2], False)])
@pytest.mark.parametrize('shape', [(4, 32, 257, 131), (2, 64, 33, 35)])
def test_upfirdn2d_up_layer_fir_register_tiled(dev, pad, flip, shape):
    """The StyleGAN2 up-layer FIR (up = down = 1, 4x4 taps, channels_last) on the sliding-window kernel, both tile shapes
    (4 columns x 16 rows / 2 x 4), ragged right / bottom edges, the backward's pads and flipped taps, cropping pads, and the fused
    bias + lrelu epilogue — against the golden-pinned oracle."""
    from layoutdetr_amd.torch_utils.ops import upfirdn2d
    torch.manual_seed(5)
    x = torch.randn(*shape)
    f = ops_ref.setup_filter([1, 3, 3, 1]) + 0.01 * torch.arange(16.).reshape(4, 4)
    xr = x.clone().requires_grad_(True)
    yr = ops_ref.upfirdn2d(xr, f, padding=pad, gain=4.0, flip_filter=flip)
    g = torch.randn_like(yr); yr.backward(g)
    xg = x.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = upfirdn2d.upfirdn2d(xg, f.to(dev), padding=pad, gain=4.0, flip_filter=flip)
    assert y.shape == yr.shape and y.is_contiguous(memory_format=torch.channels_last)
    y.backward(g.to(dev))
    assert_close(y, yr, 3e-6, 'y'); assert_close(xg.grad, xr.grad, 3e-6, 'dx')
    b = torch.randn(shape[1])
    ya = upfirdn2d._kernel_call(xg.detach(), f.to(dev), 1, 1, 1, 1, *pad, flip, 4.0, act_bias=b.to(dev), act=(0.2, 2 ** 0.5))
    assert_close(ya, torch.nn.functional.leaky_relu(yr.detach() + b.view(1, -1, 1, 1), 0.2) * 2 ** 0.5, 3e-6, 'fused bias+lrelu')


def test_upfirdn2d_separable_and_helpers(dev):
    from layoutdetr_amd.torch_utils.ops import upfirdn2d
    torch.manual_seed(3)
    x = torch.randn(1, 3, 16, 16)
    f1 = torch.tensor([1., 2., 4., 7., 7., 4., 2., 1.]); f1 = f1 / f1.sum()
    y = upfirdn2d.upfirdn2d(x.to(dev), f1.to(dev), up=2, padding=3)
    yr = ops_ref.upfirdn2d(x, f1, up=2, padding=3)
    assert_close(y, yr, 3e-6, 'separable')
    f = upfirdn2d.setup_filter([1, 3, 3, 1])
    assert torch.allclose(f, ops_ref.setup_filter([1, 3, 3, 1]))
    y = upfirdn2d.upsample2d(x.to(dev), f.to(dev)); yr = ops_ref.upsample2d(x, f)
    assert y.shape == (1, 3, 32, 32); assert_close(y, yr, 3e-6, 'upsample2d')
    with pytest.raises(RuntimeError):
        upfirdn2d.upfirdn2d(x.to(dev), f.to(dev), padding=-9)


# ------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize('M,N,K', [(18, 256, 256), (128, 2048, 256), (1024, 256, 2048), (70, 36, 52), (5, 1, 256), (300, 200, 147)])
def test_gemm_variants(dev, M, N, K):
    from layoutdetr_amd.hip import core
    torch.manual_seed(4)
    A = torch.randn(M, K); W = torch.randn(N, K); Bt = torch.randn(K, N); At = torch.randn(K, M)
    c = core.gemm(A.to(dev), W.to(dev), 0, 0, M, N, K)
    assert_close(c, A @ W.t(), 2e-6, 'NT')
    c = core.gemm(A.to(dev), Bt.to(dev), 0, 1, M, N, K)
    assert_close(c, A @ Bt, 2e-6, 'NN')
    c = core.gemm(At.to(dev), Bt.to(dev), 1, 1, M, N, K)
    assert_close(c, At.t() @ Bt, 2e-6, 'TN')
    c = core.gemm(At.to(dev), Bt.to(dev), 1, 1, M, N, K, splitk=4)
    assert_close(c, At.t() @ Bt, 2e-6, 'TN split-K')


@pytest.mark.parametrize('M,N,K', [(144, 256, 2048), (160, 256, 2048), (144, 768, 3072), (70, 36, 1100), (9, 4, 1024), (33, 65, 4100)])
def test_gemm_small_tiles_long_reduction(dev, M, N, K):
    """Few 32x32 tiles with K >= 1024 (the decoders' FFN on 9-10 tokens per sample): the small-tile kernel splits K over blocks and
    reduces through the workspace in-kernel; all operand layouts, the full epilogue, accumulate, the folded row sums, repeated
    launches (the arrival counters must re-arm), and the no-workspace route."""
    from layoutdetr_amd.hip import core
    torch.manual_seed(46)
    A = torch.randn(M, K); W = torch.randn(N, K); Bt = torch.randn(K, N); At = torch.randn(K, M)
    b = torch.randn(N); R = torch.randn(M, N)
    Ad, Wd, Btd, Atd, bd, Rd = [t.to(dev) for t in (A, W, Bt, At, b, R)]
    ref_nt = (A.double() @ W.double().t()).float()
    for _ in range(3):
        c = core.gemm(Ad, Wd, 0, 0, M, N, K, ep=core.epilogue(col_bias=bd, residual=Rd, act=core.ACT_RELU))
        assert_close(c, F.relu(ref_nt + b + R), 3e-6, 'NT + epilogue')
    assert_close(core.gemm(Ad, Btd, 0, 1, M, N, K), (A.double() @ Bt.double()).float(), 3e-6, 'NN')
    ref_tn = (At.double().t() @ Bt.double()).float()
    assert_close(core.gemm(Atd, Btd, 1, 1, M, N, K), ref_tn, 3e-6, 'TN')
    base = torch.randn(M, N); out = base.to(dev).clone()
    core.gemm(Atd, Btd, 1, 1, M, N, K, out=out, ep=core.epilogue(accumulate=True))
    assert_close(out, base + ref_tn, 3e-6, 'TN accumulate')
    rs = torch.zeros(M, device=dev)
    core.gemm(Atd, Btd, 1, 1, M, N, K, ep=core.epilogue(a_rowsum=rs))
    assert_close(rs, At.double().sum(0).float(), 3e-6, 'row sums')
    try:
        core.disable_splitk_workspace()
        assert_close(core.gemm(Ad, Wd, 0, 0, M, N, K), ref_nt, 3e-6, 'NT without workspace')
    finally:
        core.enable_splitk_workspace()


@pytest.mark.parametrize('M,N,K', [(20000, 77, 100), (4096, 256, 64), (33000, 36, 256), (8192, 300, 132), (16384, 512, 128), (4100, 1000, 32)])
def test_gemm_tall_and_thin(dev, M, N, K):
    """Dense GEMMs with M >= 4096 and K <= 256 (the trunk's 1x1 convolutions as plain matrices): both B layouts, full epilogue."""
    from layoutdetr_amd.hip import core
    torch.manual_seed(44)
    A = torch.randn(M, K); W = torch.randn(N, K); b = torch.randn(N); s = torch.rand(N) + 0.5; R = torch.randn(M, N)
    Ad, Wd, bd, sd, Rd = [t.to(dev) for t in (A, W, b, s, R)]
    ep = core.epilogue(alpha=0.5, col_scale=sd, col_bias=bd, residual=Rd, act=core.ACT_RELU)
    ref = F.relu(0.5 * (A.double() @ W.double().t()).float() * s + b + R)
    c = core.gemm(Ad, Wd, 0, 0, M, N, K, ep=ep)
    assert_close(c, ref, 3e-6, 'tall NT')
    N4 = (N + 3) // 4 * 4
    Bt = torch.randn(K, N4); Btd = Bt.to(dev)
    c = core.gemm(Ad, Btd, 0, 1, M, N4, K)
    assert_close(c, (A.double() @ Bt.double()).float(), 3e-6, 'tall NN')
    base = torch.randn(M, N4); out = base.to(dev).clone()
    core.gemm(Ad, Btd, 0, 1, M, N4, K, out=out, ep=core.epilogue(accumulate=True))
    assert_close(out, base + (A.double() @ Bt.double()).float(), 3e-6, 'tall NN accumulate')


@pytest.mark.parametrize('tokens,N,K', [(144, 256, 256), (1024, 256, 2048), (1000, 2048, 256), (4096, 1024, 512)])
def test_gemm_dw_with_bias_rowsum(dev, tokens, N, K):
    """dW = dY^T X accumulated onto an existing buffer with the bias gradient (column sums of dY) as a by-product:
    folded into the latency-bound kernel, a separate column-sum pass on the tiled path."""
    from layoutdetr_amd.hip import core
    torch.manual_seed(45)
    dY = torch.randn(tokens, N); X = torch.randn(tokens, K); gw0 = torch.randn(N, K); gb0 = torch.randn(N)
    gw = gw0.to(dev).clone(); gb = gb0.to(dev).clone(); dYd, Xd = dY.to(dev), X.to(dev)
    core.gemm(dYd, Xd, 1, 1, N, K, tokens, out=gw, ep=core.epilogue(alpha=0.5, accumulate=True, a_rowsum=gb))
    assert_close(gw, gw0 + 0.5 * (dY.double().t() @ X.double()).float(), 4e-6 * max(1.0, math.sqrt(tokens / 256)), 'dW')
    assert_close(gb, gb0 + dY.double().sum(0).float(), 1e-5, 'db')


def test_gemm_epilogue(dev):
    from layoutdetr_amd.hip import core
    torch.manual_seed(5)
    M, N, K = 96, 160, 64
    A = torch.randn(M, K); W = torch.randn(N, K); b = torch.randn(N); s = torch.rand(N) + 0.5; R = torch.randn(M, N)
    samp = torch.rand(3, N) + 0.5
    sd, bd, sampd, Rd, Ad, Wd = [t.to(dev) for t in (s, b, samp, R, A, W)]  # keep device buffers alive across launches
    ep = core.epilogue(alpha=0.5, col_scale=sd, col_bias=bd, samp_scale=sampd, residual=Rd,
                       act=core.ACT_LRELU, act_alpha=0.2, act_gain=math.sqrt(2), out_scale=1.5)
    c = core.gemm(Ad, Wd, 0, 0, M, N, K, ep=ep, pix_per_sample=32)
    ref = (A @ W.t()) * 0.5 * s * samp.repeat_interleave(32, 0) + b + R
    ref = F.leaky_relu(ref, 0.2) * math.sqrt(2) * 1.5
    assert_close(c, ref, 3e-6, 'epilogue')
    # relu-mask backward epilogue + accumulate
    Y = torch.randn(M, N)
    base = torch.randn(M, N)
    out = base.to(dev).clone()
    Yd = Y.to(dev)
    ep = core.epilogue(mask_src=Yd, mask_mode=1, accumulate=True)
    core.gemm(Ad, Wd, 0, 0, M, N, K, out=out, ep=ep)
    assert_close(out, base + (A @ W.t()) * (Y > 0), 3e-6, 'mask+accumulate')
    # dropout: keep-rate and scale
    ep = core.epilogue(p_drop=0.25, seed=1234)
    ones = torch.ones(512, 64); eye = torch.eye(64)
    d = core.gemm(ones.to(dev), eye.to(dev), 0, 0, 512, 64, 64, ep=ep).cpu()
    keep = (d != 0).float().mean().item()
    assert abs(keep - 0.75) < 0.02
    assert torch.allclose(d[d != 0], torch.tensor(1 / 0.75))


@pytest.mark.parametrize('M,N,K,sk', [(144, 256, 256, 4), (1024, 256, 2048, 3), (70, 36, 520, 5), (300, 200, 1024, 8)])
def test_gemm_splitk_fixup_and_atomic_paths(dev, M, N, K, sk):
    """Split-K through the in-kernel fix-up (workspace registered by core.lib()) and through the fp32-atomic fallback
    (workspace unregistered) with a full non-linear epilogue; the fix-up path must be bit-reproducible across launches
    and leave its arrival counters re-armed (second launch equals the first)."""
    import ctypes
    from layoutdetr_amd.hip import core
    torch.manual_seed(41)
    A = torch.randn(M, K); W = torch.randn(N, K); b = torch.randn(N); R = torch.randn(M, N)
    Ad, Wd, bd, Rd = [t.to(dev) for t in (A, W, b, R)]
    ref = F.leaky_relu(0.5 * (A.double() @ W.double().t()).float() + b + R, 0.2) * math.sqrt(2)
    ep = core.epilogue(alpha=0.5, col_bias=bd, residual=Rd, act=core.ACT_LRELU, act_alpha=0.2, act_gain=math.sqrt(2))
    tol = 4e-6 * max(1.0, math.sqrt(K / 256))
    c1 = core.gemm(Ad, Wd, 0, 0, M, N, K, splitk=sk, ep=ep)
    c2 = core.gemm(Ad, Wd, 0, 0, M, N, K, splitk=sk, ep=ep)
    assert_close(c1, ref, tol, 'fix-up split-K')
    assert torch.equal(c1, c2), 'fix-up path is not deterministic / counters not re-armed'
    ws = core._workspace[torch.cuda.current_device()]
    torch.cuda.synchronize()
    assert int(ws[:262144].view(torch.int32).abs().sum().item()) == 0, 'arrival counters left non-zero'
    lib = core.lib()
    try:
        core.check(lib.ldetr_set_workspace(None, 0), 'unregister')
        c3 = core.gemm(Ad, Wd, 0, 0, M, N, K, splitk=sk, ep=ep)
        assert_close(c3, ref, tol, 'atomic split-K')
    finally:
        core.check(lib.ldetr_set_workspace(ctypes.c_void_p(ws.data_ptr()), ws.numel() * 4), 're-register')
    c4 = core.gemm(Ad, Wd, 0, 0, M, N, K, splitk=sk, ep=ep)
    assert torch.equal(c1, c4)


# ------------------------------------------------------------------------------------------ conv
CONV_CASES = [
    # N, H, W, Cin, Cout, k, stride, pad
    (2, 16, 16, 64, 64, 1, 1, 0),
    (2, 16, 16, 64, 128, 3, 1, 1),
    (2, 17, 15, 32, 64, 3, 2, 1),
    (2, 16, 16, 64, 128, 1, 2, 0),
    (3, 9, 9, 8, 12, 3, 1, 1),
    (1, 8, 8, 256, 64, 3, 1, 1),
    # ResNet-50 stage transitions at 64x64 / 256x256 inputs
    (2, 8, 8, 256, 256, 3, 2, 1),
    (2, 16, 16, 128, 128, 3, 2, 1),
    (2, 4, 4, 512, 512, 3, 2, 1),
    (2, 8, 8, 512, 1024, 1, 2, 0),
    (2, 8, 8, 512, 256, 1, 1, 0),
    (2, 4, 4, 256, 1024, 1, 1, 0),
    (2, 32, 32, 256, 128, 1, 1, 0),
    # 1x1 / stride 1 with >= 4096 pixels and K <= 256 (plain tall-and-thin matrices for the engine), fwd and data gradient,
    # incl. ragged pixel counts, channel counts that are not multiples of 32 and every K bucket (<=64, <=128, <=256)
    (4, 32, 32, 64, 256, 1, 1, 0),
    (4, 32, 32, 256, 64, 1, 1, 0),
    (2, 47, 49, 72, 200, 1, 1, 0),
    (2, 48, 48, 128, 36, 1, 1, 0),
    (1, 65, 65, 40, 260, 1, 1, 0),
    (2, 64, 64, 192, 96, 1, 1, 0),
    # scalar-addressed (FAST) instantiation and its boundaries: channel counts of 2+ k-tiles, 5x5 (25 taps fit the tap mask) vs 7x7
    # (49 do not: generic path), stride 2 with even / odd sizes, image rows that are multiples / divisors / neither of the 32-pixel
    # k-tile (the weight gradient's pixel-major operand), a pad larger than 1, and channel counts that are not k-tile multiples
    (2, 32, 32, 64, 64, 3, 1, 1),
    (2, 64, 32, 128, 64, 3, 1, 1),
    (1, 20, 24, 64, 96, 3, 1, 1),
    (2, 32, 32, 64, 64, 5, 1, 2),
    (1, 16, 16, 64, 64, 7, 1, 3),
    (2, 33, 31, 128, 64, 3, 2, 1),
    (2, 32, 32, 96, 64, 3, 2, 1),
    (3, 8, 4, 128, 128, 3, 1, 1),
    (2, 16, 16, 80, 64, 3, 1, 1),
    (4, 256, 256, 32, 32, 3, 1, 1),      # 32 -> 32 channels on >= 2^18 pixels: wgrad_c32_3x3_kernel (no LDS, operands streamed straight into the MFMAs)
    # narrow outputs on many pixels: the 256x32 tile (Cout <= 32 forward, Cin <= 32 data gradient), ragged last tile, stride 2, 24 channels
    (2, 256, 257, 32, 32, 3, 1, 1),
    (3, 212, 208, 64, 32, 3, 1, 1),
    (2, 300, 300, 32, 64, 3, 2, 1),
    (2, 256, 256, 24, 24, 3, 1, 1),
]


@pytest.mark.parametrize('case', CONV_CASES)
def test_conv2d_fwd_bwd(dev, case):
    from layoutdetr_amd.hip import conv
    N, H, W, Ci, Co, k, s, p = case
    torch.manual_seed(6)
    x = torch.randn(N, Ci, H, W); w = torch.randn(Co, Ci, k, k) / math.sqrt(Ci * k * k)
    scale = torch.rand(Co) + 0.5; shift = torch.randn(Co)
    xr = x.clone().requires_grad_(True); wr = w.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, stride=s, padding=p) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    res = torch.randn_like(yr); rr = res.clone().requires_grad_(True)
    # ReLU only on the small cases: among millions of outputs some pre-activation lies within rounding distance of 0, its mask flips
    # between two fp32 evaluations and moves dx around that pixel by percents (seen: 1.7e-2 on a 4 x 32 x 256 x 256 output)
    relu = yr.numel() <= 2_000_000
    yr = F.relu(yr + rr) if relu else yr + rr
    g = torch.randn_like(yr); yr.backward(g)
    xg = x.permute(0, 2, 3, 1).contiguous().to(dev).requires_grad_(True)
    wg = w.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    rg = res.permute(0, 2, 3, 1).contiguous().to(dev).requires_grad_(True)
    y = conv.conv2d_nhwc(xg, wg, scale.to(dev), shift.to(dev), rg, stride=s, pad=p, relu=relu)
    y.backward(g.permute(0, 2, 3, 1).contiguous().to(dev))
    assert_close(y.permute(0, 3, 1, 2), yr, 3e-6, 'y')
    assert_close(xg.grad.permute(0, 3, 1, 2), xr.grad, 5e-6, 'dx')
    assert_close(wg.grad, wr.grad, 1e-5, 'dw')
    assert_close(rg.grad.permute(0, 3, 1, 2), rr.grad, 3e-6, 'dres')


@pytest.mark.parametrize('case', [(2, 16, 16, 64, 32, 3, 1), (2, 17, 15, 32, 64, 3, 2), (4, 32, 32, 64, 128, 1, 1), (2, 8, 8, 256, 512, 1, 1)])
def test_conv_relu_gradient_handoff(dev, case):
    """conv_a -> ReLU -> conv_b where conv_b applies the (input > 0) mask in its data-gradient epilogue and conv_a skips its
    activation-gradient pass (Bottleneck conv1->conv2->conv3): gradients equal the ordinary chain's."""
    from layoutdetr_amd.hip import conv
    N, H, W, Ca, Cb, k, s = case
    torch.manual_seed(61)
    x = torch.randn(N, 24, H, W); wa = torch.randn(Ca, 24, 1, 1) / math.sqrt(24); wb = torch.randn(Cb, Ca, k, k) / math.sqrt(Ca * k * k)
    sa = torch.rand(Ca) + 0.5; ba = torch.randn(Ca) * 0.3; sb = torch.rand(Cb) + 0.5; bb = torch.randn(Cb) * 0.3
    xr = x.clone().requires_grad_(True); war = wa.clone().requires_grad_(True); wbr = wb.clone().requires_grad_(True)
    ya = F.relu(F.conv2d(xr, war) * sa.view(1, -1, 1, 1) + ba.view(1, -1, 1, 1))
    yb = F.relu(F.conv2d(ya, wbr, stride=s, padding=k // 2) * sb.view(1, -1, 1, 1) + bb.view(1, -1, 1, 1))
    g = torch.randn_like(yb); yb.backward(g)
    xg = x.permute(0, 2, 3, 1).contiguous().to(dev).requires_grad_(True)
    wag = wa.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wbg = wb.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    a = conv.conv2d_nhwc(xg, wag, sa.to(dev), ba.to(dev), None, 1, 0, relu=True, premasked=True)
    b = conv.conv2d_nhwc(a, wbg, sb.to(dev), bb.to(dev), None, s, k // 2, relu=True, mask_input=True)
    b.backward(g.permute(0, 2, 3, 1).contiguous().to(dev))
    assert_close(b.permute(0, 3, 1, 2), yb, 3e-6, 'y')
    assert_close(wbg.grad, wbr.grad, 1e-5, 'dw_b')
    assert_close(wag.grad, war.grad, 1e-5, 'dw_a')
    assert_close(xg.grad.permute(0, 3, 1, 2), xr.grad, 5e-6, 'dx')


@pytest.mark.parametrize('N,H,W', [(2, 32, 32), (3, 50, 76), (1, 37, 131), (2, 256, 256)])
def test_conv2d_stem_nchw(dev, N, H, W):
    """ResNet stem 7x7 / 2 on an NCHW image (csrc/stem_conv.hip: LDS-resident patch + weights) with FrozenBN + ReLU fused; sizes that do
    not fill the 8 x 32 output tiles, odd sizes; the weight gradient (engine, scalar-gather operand view) alongside."""
    from layoutdetr_amd.hip import conv
    torch.manual_seed(7)
    x = torch.randn(N, 3, H, W); w = torch.randn(64, 3, 7, 7) * 0.1
    scale = torch.rand(64) + 0.5; shift = torch.randn(64)
    wg = w.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = conv.conv2d_nhwc(x.to(dev), wg, scale.to(dev), shift.to(dev), None, stride=2, pad=3, relu=True, x_is_nchw=True)
    wr = w.clone().requires_grad_(True)
    pre = F.conv2d(x, wr, stride=2, padding=3) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    # the ReLU mask of the reference backward is the device's: among 10^5..10^6 outputs a few pre-activations lie within rounding
    # distance of zero, and one flipped unit moves dw by ~1e-3 of its maximum (checked below: disagreements only there)
    mask = (y.permute(0, 3, 1, 2) > 0).cpu()
    flips = mask != (pre.detach() > 0)
    assert flips.sum().item() <= 8 and (pre.detach()[flips].abs() < 1e-5).all()
    yr = pre * mask
    g = torch.randn_like(yr); yr.backward(g)
    y.backward(g.permute(0, 2, 3, 1).contiguous().to(dev))
    assert_close(y.permute(0, 3, 1, 2), yr.detach(), 3e-6, 'stem y'); assert_close(wg.grad, wr.grad, 1e-5, 'stem dw')


def test_maxpool(dev):
    from layoutdetr_amd.hip import conv
    torch.manual_seed(8)
    x = F.relu(torch.randn(2, 8, 13, 12))  # ties at 0 exercise first-max routing
    xr = x.clone().requires_grad_(True)
    yr = F.max_pool2d(xr, 3, 2, 1); g = torch.randn_like(yr); yr.backward(g)
    xg = x.permute(0, 2, 3, 1).contiguous().to(dev).requires_grad_(True)
    y = conv.maxpool3x3s2_nhwc(xg); y.backward(g.permute(0, 2, 3, 1).contiguous().to(dev))
    assert torch.equal(y.permute(0, 3, 1, 2).cpu(), yr.detach())
    assert_close(xg.grad.permute(0, 3, 1, 2), xr.grad, 1e-6, 'maxpool dx')


# ------------------------------------------------------------------------------------------ attention
@pytest.mark.parametrize('B,H,Lq,Lk,dh', [(2, 8, 9, 9, 32), (2, 8, 10, 64, 32), (3, 2, 64, 64, 32), (2, 8, 33, 40, 32), (2, 3, 64, 17, 32), (1, 4, 16, 256, 32), (2, 8, 33, 100, 32),
                                          (2, 8, 9, 257, 32), (2, 2, 300, 400, 32), (1, 4, 16, 1024, 32), (1, 2, 1024, 1024, 32), (2, 1, 40, 2100, 32),
                                          (2, 4, 9, 64, 64), (2, 2, 10, 256, 128), (2, 4, 70, 300, 64), (2, 3, 9, 9, 96)])
def test_attention(dev, B, H, Lq, Lk, dh):
    """Lk <= 256: scores in registers; above: 256-key chunks with an online softmax (background_size 1024 -> 1024 memory tokens).
    32-wide heads with 17..64 keys and <= 64 queries take the LDS-staged backward (attn_bwd_lds_kernel), 48+ queries the LDS-staged forward.
    Head widths 32 (the DETR blocks at 8 heads) .. 128 (hidden 256 at 4 / 2 heads), cross-attention shapes (Lq != Lk)."""
    from layoutdetr_amd.hip import attention
    torch.manual_seed(9)
    d = H * dh
    q = torch.randn(B * Lq, d); k = torch.randn(B * Lk, d); v = torch.randn(B * Lk, d)
    kpm = torch.zeros(B, Lk, dtype=torch.bool)
    kpm[0, Lk - Lk // 3:] = True
    if Lk > 256:
        kpm[-1, :300 if Lk > 300 else 256] = True     # a whole first chunk masked: the running maximum starts at -inf
    qr, kr, vr = [t.clone().requires_grad_(True) for t in (q, k, v)]

    def heads(t, L):
        return t.view(B, L, H, dh).permute(0, 2, 1, 3)
    s = heads(qr, Lq) @ heads(kr, Lk).transpose(-1, -2) / math.sqrt(dh)
    s = s.masked_fill(kpm[:, None, None, :], float('-inf'))
    o = (s.softmax(-1) @ heads(vr, Lk)).permute(0, 2, 1, 3).reshape(B * Lq, d)
    g = torch.randn_like(o); o.backward(g)
    qg, kg, vg = [t.to(dev).requires_grad_(True) for t in (q, k, v)]
    og = attention.attention(qg, kg, vg, kpm.to(dev), B, H, Lq, Lk, 0.0)
    og.backward(g.to(dev))
    assert_close(og, o, 5e-6, 'o'); assert_close(qg.grad, qr.grad, 2e-5, 'dq')
    assert_close(kg.grad, kr.grad, 2e-5, 'dk'); assert_close(vg.grad, vr.grad, 2e-5, 'dv')


def test_attention_dropout_statistics(dev):
    from layoutdetr_amd.hip import attention
    torch.manual_seed(10)
    B, H, L = 4, 8, 64
    q = torch.zeros(B * L, 256, device=dev); k = torch.zeros(B * L, 256, device=dev)
    v = torch.ones(B * L, 256, device=dev).requires_grad_(True)
    o = attention.attention(q, k, v, None, B, H, L, L, 0.1)
    # uniform probabilities 1/L, kept w.p. 0.9 and rescaled by 1/0.9  ->  E[o] = 1
    assert abs(o.mean().item() - 1.0) < 0.01 and o.std().item() > 1e-3
    o.sum().backward()  # mask is regenerated in backward: dV column sums equal the forward keep pattern
    assert abs(v.grad.mean().item() - 1.0) < 0.01
    # long-key path (chunked): same statistics, and the backward regenerates the same mask (dV mean equals the forward's keep rate)
    L2 = 512
    q = torch.zeros(2 * 16, 256, device=dev); k = torch.zeros(2 * L2, 256, device=dev)
    v = torch.ones(2 * L2, 256, device=dev).requires_grad_(True)
    o = attention.attention(q, k, v, None, 2, 8, 16, L2, 0.1)
    assert abs(o.mean().item() - 1.0) < 0.01 and o.std().item() > 1e-3
    o.sum().backward()
    assert abs(v.grad.sum().item() / (16 * 2 * 256) - o.mean().item()) < 1e-4


# ------------------------------------------------------------------------------------------ layernorm
@pytest.mark.parametrize('rows,D', [(18, 256), (1024, 256), (37, 768), (5, 64)])
def test_layernorm(dev, rows, D):
    from layoutdetr_amd.hip import layernorm
    torch.manual_seed(11)
    x = torch.randn(rows, D); r = torch.randn(rows, D); gm = torch.rand(D) + 0.5; bt = torch.randn(D)
    xr, rr, gr, br = [t.clone().requires_grad_(True) for t in (x, r, gm, bt)]
    yr = F.layer_norm(xr + rr, (D,), gr, br, 1e-5); g = torch.randn_like(yr); yr.backward(g)
    xg, rg, gg, bg = [t.to(dev).requires_grad_(True) for t in (x, r, gm, bt)]
    y = layernorm.add_layernorm(xg, rg, gg, bg, 1e-5, 0.0); y.backward(g.to(dev))
    assert_close(y, yr, 3e-6, 'y'); assert_close(xg.grad, xr.grad, 1e-5, 'dx'); assert_close(rg.grad, rr.grad, 1e-5, 'dr')
    assert_close(gg.grad, gr.grad, 1e-5, 'dgamma'); assert_close(bg.grad, br.grad, 1e-5, 'dbeta')
    y2 = layernorm.add_layernorm(xg.detach(), None, gg.detach(), bg.detach())
    assert_close(y2, F.layer_norm(x, (D,), gm, bt, 1e-5), 3e-6, 'plain LN')


# ------------------------------------------------------------------------------------------ linear
def test_linear_autograd(dev):
    from layoutdetr_amd.hip import linear, core
    torch.manual_seed(12)
    x = torch.randn(4, 9, 256); w = torch.randn(2048, 256) * 0.05; b = torch.randn(2048)
    xr, wr, br = [t.clone().requires_grad_(True) for t in (x, w, b)]
    yr = F.relu(F.linear(xr, wr, br)); g = torch.randn_like(yr); yr.backward(g)
    xg, wg, bg = [t.to(dev).requires_grad_(True) for t in (x, w, b)]
    y = linear.linear(xg, wg, bg, act=core.ACT_RELU); y.backward(g.to(dev))
    assert_close(y, yr, 3e-6, 'y'); assert_close(xg.grad, xr.grad, 5e-6, 'dx')
    assert_close(wg.grad, wr.grad, 5e-6, 'dw'); assert_close(bg.grad, br.grad, 5e-6, 'db')


def test_split_bf16_non_finite_operands_match_f32_pipe_and_reference_sanitiser(dev):
    """The exact three-way split of +-Inf (and of a finite value within one bf16 ulp of FLT_MAX) is Inf + NaN + NaN: without care the bf16
    pipe would return NaN where the f32 pipe (and the reference's fp32 GEMM) returns +-Inf, and the step's gradient sanitiser
    (training_loop.py:306-309: nan -> 0, +inf -> 1e5, -inf -> -1e5) maps the two differently.  A tile whose accumulators come out
    non-finite is recomputed on the f32 pipe inside the launch: (a) Inf / NaN / finite CLASSES of every output equal the f32 pipe's and
    torch's fp32 matmul's, finite values agree; (b) fed through the fused sanitise + Adam kernel the parameters equal the reference
    post-processing (oracle losses_ref.dp_postprocess, golden-pinned) + torch Adam."""
    from layoutdetr_amd.hip import core
    from oracle import losses_ref
    torch.manual_seed(44)
    for ta, tb, M, N, K in [(0, 0, 4096, 512, 1152), (1, 1, 1024, 1024, 8192), (0, 1, 16384, 256, 512)]:      # (shapes whose tiles run on the split pipe)
        A = torch.randn((K, M) if ta else (M, K)); B = torch.randn((K, N) if tb else (N, K)).abs() + 0.1      # B > 0: no Inf - Inf in a row
        Av = A.t() if ta else A                                      # logical [M, K] view of the storage
        Av[3, 5] = float('inf'); Av[70, 9] = -float('inf'); Av[200, 1] = float('nan'); Av[333, 7] = 3.4e38   # 3.4e38: bf16(hi) rounds to Inf
        Av[400, 2] = float('inf'); Av[400, 3] = -float('inf')                                            # Inf - Inf -> NaN on every pipe
        ref = Av @ (B if tb else B.t())                              # torch fp32 on the host
        Ad, Bd = A.to(dev), B.to(dev)
        f32, sp = _both_pipes(lambda: core.gemm(Ad, Bd, ta, tb, M, N, K).clone())
        for name, out in (('f32 pipe', f32.cpu()), ('bf16 split', sp.cpu())):
            assert torch.equal(torch.isnan(out), torch.isnan(ref)), f'{name} ta={ta} tb={tb}: NaN positions differ from the fp32 matmul'
            assert torch.equal(torch.isposinf(out), torch.isposinf(ref)) and torch.equal(torch.isneginf(out), torch.isneginf(ref)), f'{name}: Inf positions differ'
            clean = torch.ones(M, dtype=torch.bool); clean[[3, 70, 200, 333, 400]] = False      # rows of the SAME tiles as the poisoned ones included
            assert (out[clean] - ref[clean]).abs().max() <= 2e-5 * ref[clean].abs().max(), f'{name}: finite rows of a recomputed tile'
            big = torch.isfinite(ref[333])                                                      # the 3.4e38 row: finite where the products stay below FLT_MAX
            assert ((out[333][big] - ref[333][big]).abs() <= 1e-5 * ref[333][big].abs()).all(), name
        assert torch.isposinf(sp[3]).all() and torch.isneginf(sp[70]).all() and torch.isnan(sp[200]).all() and torch.isnan(sp[400]).all()
        assert not torch.equal(f32[500:], sp[500:]), 'both runs took the same path: the split tiles were not exercised'
        # (b) such a product as a weight gradient through `/world` + nan_to_num + Adam (fuse_sanitize = 1)
        n = sp.numel(); world = 2
        g_ref = losses_ref.dp_postprocess(ref.flatten().clone() * 1.0, world)
        p0 = torch.randn(n)
        pr = p0.clone().requires_grad_(True); opt = torch.optim.Adam([pr], lr=1e-3, betas=(0.0, 0.99), eps=1e-8)
        pr.grad = g_ref.clone(); opt.step()
        pg = p0.to(dev); m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev); gd = sp.flatten().contiguous()
        core.check(core.lib().ldetr_adam_step_f32(core.ptr(pg), core.ptr(gd), core.ptr(m), core.ptr(v), n, 1, 1e-3, 0.0, 0.99, 1e-8,
                                                  1, 1.0 / world, 0.0, 1e5, -1e5, core.stream()))
        mc = m.cpu()
        assert torch.equal(mc == 1e5, g_ref == 1e5) and torch.equal(mc == -1e5, g_ref == -1e5) and (mc == 1e5).any() and (mc == -1e5).any()
        assert torch.equal((mc == 0) & ~torch.isfinite(ref.flatten()), torch.isnan(ref.flatten()))
        fin = g_ref.abs() < 1e4
        assert (mc[fin] - g_ref[fin]).abs().max() <= 2e-5 * g_ref[fin].abs().max(), 'first moment (= sanitised gradient at beta1 = 0)'
        big = g_ref.abs() > 1e-2 * g_ref[fin].abs().max()       # (Adam's first step is lr * g / (|g| + eps): only well away from 0 is it insensitive to the last bits of g)
        assert_close(pg.cpu()[big], pr.detach()[big], 1e-5, 'adam after sanitising a split-pipe gradient')


# ------------------------------------------------------------------------------------------ loss tails (hip/losses.py)
@pytest.mark.parametrize('B', [1, 2, 16, 300])
def test_loss_combine_and_masked_mse_vs_torch_autograd(dev, B):
    """hip.losses.combine (softplus(+-x) per-sample terms, per-sample shares of scalar terms, scalars, the cross-entropy (sum, count) pair; batch mean,
    gain) and hip.losses.masked_mse against the reference's op chain (training/loss.py:195-213: F.softplus, weights, sum, .mean().mul(gain)) in
    float64 autograd: total, every reported term and every gradient."""
    from layoutdetr_amd.hip import losses as hl
    torch.manual_seed(700 + B)
    mk = lambda *s: torch.randn(*s, dtype=torch.float64)
    logit, logit2 = mk(B) * 3, mk(B) * 30                    # (|x| > 20: softplus' threshold branch)
    lay, sc, ce = mk(4, B).abs(), mk(()), torch.tensor([7.5, 3.0], dtype=torch.float64)
    leaves = [t.clone().requires_grad_(True) for t in (logit, logit2, lay, sc, ce)]
    gain, W = 0.5, [1.0, 1.0, 100.0, 4.0, 7.0, 17.0, 5.0, 50.0]
    a, b, l4, s0, c2 = leaves
    ref = {'p': F.softplus(-a) * W[0], 'q': F.softplus(b) * W[1], 'r': l4[0].sum() * W[2], 'g': l4[1].sum() * W[3], 'o': l4[2] * W[4], 'al': l4[3] * W[5],
           's': s0 * W[6], 'c': c2[0] / c2[1] * W[7]}
    tot_ref = sum(ref.values()).mean() * gain
    tot_ref.backward()
    gl = [t.detach().float().to(dev).requires_grad_(True) for t in (logit, logit2, lay, sc, ce)]
    T = hl.Term
    total, rep = hl.combine([T('p', gl[0], W[0], hl.SOFTPLUS_NEG), T('q', gl[1], W[1], hl.SOFTPLUS), T(['r', 'g', 'o', 'al'], gl[2], W[2:6], hl.IDENT, [True, True, False, False]),
                             T('s', gl[3], W[6]), T('c', gl[4], W[7], hl.RATIO)], gain)
    total.backward()
    assert_close(total, tot_ref, 2e-6, 'total')
    for k, v in ref.items():
        assert_close(rep[k], v.detach(), 2e-6, 'term ' + k)
    for name, g, r in zip(('d logit', 'd logit2', 'd layout', 'd scalar'), gl[:4], leaves[:4]):
        assert_close(g.grad, r.grad, 2e-6, name)
    # the ratio's gradient leaves undivided (the cross-entropy backward divides by its count): d total / d (sum / count) * 1
    assert_close(gl[4].grad[0], leaves[4].grad[0] * ce[1], 2e-6, 'd ce sum'); assert float(gl[4].grad[1]) == 0.0
    # masked mse: per-slot reference rows and one reference row per sample
    N, D = 9, 36
    x = mk(B, N, D); y = mk(B, N, D); z = mk(B, D); valid = torch.rand(B, N) > 0.3
    if B == 1:
        valid[:] = False                     # no valid slot at all: 0 / max(0, 1), zero gradient
    for bref, bdiv in ((y, 1), (z, N)):
        xr = x.clone().requires_grad_(True)
        full = bref if bdiv == 1 else bref.unsqueeze(1).expand(-1, N, -1)
        if valid.any():
            lr = F.mse_loss(xr[valid], full[valid]); lr.backward()
        else:
            lr = torch.zeros((), dtype=torch.float64); xr.grad = torch.zeros_like(x)
        xg = x.float().to(dev).requires_grad_(True)
        lg = hl.masked_mse(xg, bref.float().to(dev), valid.to(dev).to(torch.uint8), bdiv=bdiv)
        (lg * 3.0).backward()
        assert_close(lg, lr.detach(), 2e-6, 'masked mse'); assert_close(xg.grad, xr.grad * 3.0, 2e-6, 'd masked mse')


# ------------------------------------------------------------------------------------------ the short token stacks as one node (hip/stacks.py)
def _randomise(mod, seed):
    g = torch.Generator().manual_seed(seed)
    for n, p_ in mod.named_parameters():
        if n.endswith('bias'):
            p_.data = torch.randn(p_.shape, generator=g) * 0.2
        elif 'norm' in n and n.endswith('weight'):
            p_.data = torch.rand(p_.shape, generator=g) + 0.5


@pytest.mark.parametrize('kind,B,L,S,masked', [('enc', 16, 9, 0, True), ('enc', 5, 10, 0, True), ('enc', 3, 16, 0, False), ('enc', 1, 1, 0, False),
                                               ('dec', 16, 10, 64, True), ('dec', 4, 9, 4, True), ('dec', 3, 16, 37, True), ('dec', 2, 1, 64, False),
                                               ('dec', 3, 9, 80, True)])
def test_token_stack_node_vs_fp64_oracle(dev, kind, B, L, S, masked):
    """A 2-layer encoder-type stack (nn.TransformerEncoderLayer semantics, training/util.py:13-43) and a 2-layer DETR decoder stack + final norm
    (training/detr_transformer.py:265-286, 88) at d_model 256 / 8 heads / 2048 hidden through the stack node (hip/stacks.py: one-launch self- and
    cross-attention sub-blocks forward AND backward, fused feed-forward block, partial-sum LayerNorms, one weight-gradient launch per layer) against
    the fixture-pinned oracle (oracle/detr_ref.py) evaluated in float64: output, input gradient, the projected memory's gradients and EVERY parameter
    gradient, entry by entry; ragged key-padding masks; S = 80 memory tokens takes the node's generic cross-attention path (> 64 keys)."""
    from layoutdetr_amd.hip import stacks as hstacks
    from layoutdetr_amd.training import detr_transformer as T
    from oracle import detr_ref
    torch.manual_seed(300 + 7 * B + L + S)
    d, H = 256, 8
    if kind == 'enc':
        mod = T.TransformerEncoder(T.TransformerEncoderLayer(d_model=d, nhead=H, dim_feedforward=2048), num_layers=2)
    else:
        mod = T.TransformerDecoder(T.TransformerDecoderLayer(d_model=d, nhead=H, dim_feedforward=2048), num_layers=2, norm=torch.nn.LayerNorm(d))
    _randomise(mod, 17)
    mod.eval()
    x = torch.randn(B * L, d); gy = torch.randn(B * L, d)
    kpm = torch.zeros(B, L, dtype=torch.bool)
    mem_kpm = torch.zeros(B, max(S, 1), dtype=torch.bool)
    if masked:
        kpm[0, L - L // 3:] = True
        if B > 2:
            kpm[2, 1:] = True
        if S:
            mem_kpm[1 % B, S - S // 3:] = True
    mem = torch.randn(B * S, d) if S else None
    pos = torch.randn(S, d) * 0.3 if S else None
    # ---- float64 oracle (seq-first tensors)
    sd = {k: v.detach().double().requires_grad_(True) for k, v in mod.state_dict().items()}
    xr = x.double().requires_grad_(True)
    sf = lambda t, n: t.reshape(B, n, d).permute(1, 0, 2)
    if kind == 'enc':
        yr = detr_ref.torch_encoder(sd, '', sf(xr, L), H, kpm)
    else:
        memr = mem.double().requires_grad_(True)
        y_ = sf(xr, L)
        for i in range(2):
            y_ = detr_ref.decoder_layer(sd, f'layers.{i}.', y_, sf(memr, S), H, kpm, mem_kpm, pos.double()[:, None, :].expand(S, B, d))
        yr = detr_ref._ln(sd, 'norm.', y_)
    yr = yr.permute(1, 0, 2).reshape(B * L, d)
    (yr * gy.double()).sum().backward()
    # ---- HIP
    mod.to(dev)
    xg = x.to(dev).requires_grad_(True)
    assert hstacks.ENABLED
    n0 = hstacks.NODE_RUNS[0]
    if kind == 'enc':
        y = mod.forward2d(xg, B, L, kpm.to(dev), None)
    else:
        memg = mem.to(dev).requires_grad_(True)
        pos2 = pos.to(dev).repeat(B, 1)
        y = mod.forward2d(xg, memg, pos2, B, L, S, kpm.to(dev), mem_kpm.to(dev))
    assert hstacks.NODE_RUNS[0] == n0 + 1, 'the stack did not take the stack node'
    (y * gy.to(dev)).sum().backward()
    assert_close(y, yr, 2e-5, 'y')
    assert_close(xg.grad, xr.grad, 5e-5, 'dx')
    if kind == 'dec':
        assert_close(memg.grad, memr.grad, 5e-5, 'd memory')
    named = dict(mod.named_parameters())
    for k, v in sd.items():
        if v.grad is None:
            continue
        assert named[k].grad is not None, k
        assert_close(named[k].grad, v.grad, 5e-5, 'd ' + k)


def test_token_stacks_in_a_group_equal_the_stacks_alone_bit_for_bit(dev):
    """Two independent stacks in lock-step (ONE launch per sub-block step for both: the second problem's blocks follow the first's in the grid) give
    exactly the values and gradients of the two stacks run one after the other -- an encoder-type pair (D's reconstruction decoders) and a decoder
    beside an encoder-type stack (D's layout decoder beside its unconditional encoder), dropout on (same seeds in both runs), gradients accumulated
    into a FlatModule buffer on top of existing content."""
    from layoutdetr_amd.hip import core
    from layoutdetr_amd.hip import stacks as hstacks
    from layoutdetr_amd.hip.attention import grouped_kv
    from layoutdetr_amd.training import detr_transformer as T
    from layoutdetr_amd.training.training_loop import FlatModule
    torch.manual_seed(411)
    d, H, B, L, S = 256, 8, 16, 9, 64
    ea = T.TransformerEncoder(T.TransformerEncoderLayer(d_model=d, nhead=H, dim_feedforward=2048), num_layers=3)
    eb = T.TransformerEncoder(T.TransformerEncoderLayer(d_model=d, nhead=H, dim_feedforward=2048), num_layers=3)
    dc = T.TransformerDecoder(T.TransformerDecoderLayer(d_model=d, nhead=H, dim_feedforward=2048), num_layers=3, norm=torch.nn.LayerNorm(d))
    mods = torch.nn.ModuleList([ea, eb, dc]).to(dev).train()
    fm = FlatModule(mods)
    xa0, xb0, xd0 = torch.randn(B * L, d, device=dev), torch.randn(B * L, d, device=dev), torch.randn(B * L, d, device=dev)
    mem0 = torch.randn(B * S, d, device=dev); pos2 = (torch.randn(S, d, device=dev) * 0.3).repeat(B, 1)
    kpm = torch.zeros(B, L, dtype=torch.uint8, device=dev); kpm[2, 4:] = 1; kpm[7, 1:] = 1
    ga, gb, gd = torch.randn(B * L, d, device=dev), torch.randn(B * L, d, device=dev), torch.randn(B * L, d, device=dev)
    res = {}
    for grouped in (False, True):
        fm.zero_grad(); fm.gflat.fill_(0.125)
        xa, xb, xd, mem = (t.clone().requires_grad_(True) for t in (xa0, xb0, xd0, mem0))
        core._seed_counter[0] = 5000
        pa, pb = ea.as_prog(xa, B, L, kpm), eb.as_prog(xb, B, L, kpm)
        kvs = grouped_kv(mem + pos2, mem, [l.multihead_attn for l in dc.layers])
        pd = hstacks.Prog('dec', dc.layers, xd, B, L, kpm, True, final_norm=dc.norm, kvs=kvs, S=S, mem_kpm=None)
        pe = eb.as_prog(xa * 0.5, B, L, kpm)           # (a second use of eb's weights: beside the decoder)
        assert pa is not None and pb is not None and pe is not None and hstacks.usable(pd)
        if grouped:
            ya, yb = hstacks.run([pa, pb])
            yd, ye = hstacks.run([pd, pe])
        else:
            # the same seed order as the grouped run: per layer step the first stack's draws, then the second's
            ya, yb, yd, ye = _run_alone_with_group_seed_order(hstacks, core, [pa, pb], [pd, pe])
        ((ya * ga).sum() + (yb * gb).sum() + (yd * gd).sum() + (ye * ga).sum()).backward()
        res[grouped] = dict(ya=ya.detach().clone(), yb=yb.detach().clone(), yd=yd.detach().clone(), ye=ye.detach().clone(), dxa=xa.grad.clone(), dxb=xb.grad.clone(),
                            dxd=xd.grad.clone(), dmem=mem.grad.clone(), g=fm.gflat.clone())
    for k, v in res[False].items():
        if k == 'g':      # weight gradients of eb arrive from two nodes: their order in the flat buffer's += is autograd's; values to rounding
            assert_close(res[True][k], v, 1e-6, k)
        else:
            assert torch.equal(res[True][k], v), f'{k}: grouped launch differs from the single launches ({(res[True][k] - v).abs().max().item():.3e})'


def _run_alone_with_group_seed_order(hstacks, core, *groups):
    """Each stack of a group as its own node, with the dropout seeds the grouped run would have drawn.  hip.core.next_seed is a counter: a group draws
    per layer step across its stacks (every stack's attention seed; a decoder's norm1 + cross-attention seeds; every stack's norm_a, hidden, norm_b
    seeds), a lone stack layer by layer on its own -- so the lone runs are handed the group's draw indices."""
    outs = []
    for progs in groups:
        base, real = core._seed_counter[0], core.next_seed
        order = []                                   # stack identity of every draw of the grouped run, in draw order
        for i in range(max(len(p.layers) for p in progs)):
            act = [p for p in progs if i < len(p.layers)]
            order += [id(p) for p in act]
            for p in act:
                if p.kind == 'dec':
                    order += [id(p), id(p)]
            for p in act:
                order += [id(p)] * 3
        for p in progs:
            mine = iter([j for j, pid in enumerate(order) if pid == id(p)])

            def fake(mine=mine):
                core._seed_counter[0] = base + next(mine)
                return real()
            core.next_seed = fake
            try:
                outs.append(hstacks.run([p])[0])
            finally:
                core.next_seed = real
        core._seed_counter[0] = base + len(order)
    return outs


# ------------------------------------------------------------------------------------------ fused feed-forward block
@pytest.mark.parametrize('M,F,pos', [(144, 2048, False), (160, 2048, True), (9, 128, False), (320, 2048, False), (33, 64, True)])
def test_ln_ffn_ln_fused_tail_vs_fp64_reference(dev, M, F, pos):
    """x1 = LN_a(x + r); y = LN_b(x1 + linear2(relu(linear1(x1)))) (dropout off) through hip.ffn (LayerNorm launch, one fused feed-forward
    launch, the partial-sum LayerNorm launch; backward: LayerNorm backward, fused feed-forward backward with per-slice partial input
    gradients, one paired weight-gradient launch, the partial-sum LayerNorm backward) against an fp64 torch evaluation: output, the second
    output y + pos, dx, dr and every parameter gradient; token counts that are not multiples of the 32-row tile included."""
    from layoutdetr_amd.hip import ffn
    torch.manual_seed(60 + M)
    D = 256
    l1 = torch.nn.Linear(D, F); l2 = torch.nn.Linear(F, D); lna = torch.nn.LayerNorm(D); lnb = torch.nn.LayerNorm(D)
    for ln in (lna, lnb):
        ln.weight.data.uniform_(0.5, 1.5); ln.bias.data.normal_(0, 0.1)
    x = torch.randn(M, D); r = torch.randn(M, D); gy = torch.randn(M, D); P = torch.randn(M // 3 if M % 3 == 0 else M, D) if pos else None
    gp = torch.randn(M, D) if pos else None
    mods = (l1, l2, lna, lnb)
    refs = (torch.nn.Linear(D, F).double(), torch.nn.Linear(F, D).double(), torch.nn.LayerNorm(D).double(), torch.nn.LayerNorm(D).double())
    for a_, b_ in zip(mods, refs):
        b_.load_state_dict({k: v.double() for k, v in a_.state_dict().items()})
    l1r, l2r, lnar, lnbr = refs
    xr, rr = x.double().requires_grad_(True), r.double().requires_grad_(True)
    x1r = lnar(xr + rr)
    yr = lnbr(x1r + l2r(torch.relu(l1r(x1r))))
    lossr = (yr * gy.double()).sum()
    if pos:
        ypr = yr + P.double().repeat(M // P.shape[0], 1)
        lossr = lossr + (ypr * gp.double()).sum()
    lossr.backward()
    for m in mods:
        m.to(dev)
    xg, rg = x.to(dev).requires_grad_(True), r.to(dev).requires_grad_(True)
    assert ffn.usable(xg, l1, l2)
    out = ffn.add_ln_ffn_add_ln(xg, rg, lna, 0.0, l1, l2, lnb, 0.0, 0.0, pos=P.to(dev) if pos else None)
    y, yp = out if pos else (out, None)
    loss = (y * gy.to(dev)).sum()
    if pos:
        loss = loss + (yp * gp.to(dev)).sum()
    loss.backward()
    assert_close(y, yr, 5e-6, 'y')
    if pos:
        assert_close(yp, ypr, 5e-6, 'y + pos')

    def close_up_to_relu_flips(a, b, what):
        # fp32 vs fp64: a hidden unit whose pre-activation is within rounding of 0 lands on the other side of the relu; that moves ONE token's
        # row of dx and ONE hidden unit's rows of the weight gradients by that unit's contribution.  Everything else agrees to 2e-5.
        e = (a.detach().double().cpu() - b).abs() / b.abs().max()
        assert (e > 2e-5).double().mean().item() <= 0.03, f'{what}: {(e > 2e-5).double().mean().item():.4f} of the entries off, max {e.max().item():.2e}'
    close_up_to_relu_flips(xg.grad, xr.grad, 'dx'); close_up_to_relu_flips(rg.grad, rr.grad, 'dr')
    for a_, b_ in zip(mods, refs):
        for (n, pa), (_, pb) in zip(a_.named_parameters(), b_.named_parameters()):
            close_up_to_relu_flips(pa.grad, pb.grad, f'grad {type(a_).__name__}.{n}')


def test_ln_ffn_ln_fused_tail_train_mode_dropout_and_reproducibility(dev):
    """Train mode (hidden dropout 0.1 inside the fused launch, residual dropouts 0.1 in the LayerNorm launches): backward consistent with
    forward -- with the masks frozen (same seeds) a central finite difference of the scalar loss matches the analytic directional
    derivative; hidden dropout changes the output; and forward AND backward are bit-reproducible run to run (no atomics on the activation
    path; weight gradients returned by autograd here, also without atomics)."""
    from layoutdetr_amd.hip import core, ffn
    torch.manual_seed(66)
    M, D, F = 144, 256, 2048
    l1 = torch.nn.Linear(D, F).to(dev); l2 = torch.nn.Linear(F, D).to(dev); lna = torch.nn.LayerNorm(D).to(dev); lnb = torch.nn.LayerNorm(D).to(dev)
    x = torch.randn(M, D, device=dev); r = torch.randn(M, D, device=dev); gy = torch.randn(M, D, device=dev); v = torch.randn(M, D, device=dev)

    def run(xx, need_grad, p_h=0.1, p_res=0.1):
        core._seed_counter[0] = 0x5EED      # same seeds -> same dropout masks in every evaluation
        for m in (l1, l2, lna, lnb):
            for p_ in m.parameters():
                p_.grad = None
        xx = xx.clone().requires_grad_(need_grad)
        y = ffn.add_ln_ffn_add_ln(xx, r, lna, p_res, l1, l2, lnb, p_h, p_res)
        loss = (y * gy).sum()
        if need_grad:
            loss.backward()
            return loss.item(), xx.grad, y.detach().clone(), [p_.grad.clone() for m in (l1, l2, lna, lnb) for p_ in m.parameters()]
        return loss.item(), None, y.detach().clone(), None
    core.reseed(dev)
    _, g, y0, w0 = run(x, True)
    _, g1, y1, w1 = run(x, True)
    # (the four linear tensors; the LayerNorm scale / shift gradients are per-block partial sums merged with fp32 atomics)
    assert torch.equal(y0, y1) and torch.equal(g, g1) and all(torch.equal(a, b) for a, b in zip(w0[:4], w1[:4])), 'not reproducible run to run'
    eps = 1e-3
    with torch.no_grad():
        lp, _, _, _ = run(x + eps * v, False); lm, _, _, _ = run(x - eps * v, False)
    fd = (lp - lm) / (2 * eps); an = (g * v).sum().item()
    assert abs(fd - an) <= 2e-2 * max(abs(an), 1.0), (fd, an)
    with torch.no_grad():
        _, _, y_h, _ = run(x, False, p_h=0.1, p_res=0.0); _, _, y_e, _ = run(x, False, p_h=0.0, p_res=0.0)
    assert (y_h - y_e).abs().max() > 1e-3, 'hidden dropout had no effect'


# ------------------------------------------------------------------------------------------ optimiser / DP step kernels
def test_adam_sanitize_ema(dev):
    from layoutdetr_amd.hip import core
    torch.manual_seed(13)
    n = 100003
    p0 = torch.randn(n); g = torch.randn(n)
    pr = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([pr], lr=1e-3, betas=(0.0, 0.99), eps=1e-8)
    pg = p0.to(dev); m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev)
    for step in range(1, 4):
        gs = g * step
        pr.grad = gs.clone(); opt.step()
        core.check(core.lib().ldetr_adam_step_f32(core.ptr(pg), core.ptr(gs.to(dev)), core.ptr(m), core.ptr(v), n, step, 1e-3,
                                                  0.0, 0.99, 1e-8, 0, 1.0, 0.0, 0.0, 0.0, core.stream()))
    assert_close(pg, pr, 1e-6, 'adam')
    gbad = torch.tensor([float('nan'), float('inf'), -float('inf'), 2.0, -4.0] * 3, device=dev)
    core.check(core.lib().ldetr_grad_sanitize_f32(core.ptr(gbad), gbad.numel(), 0.5, 0.0, 1e5, -1e5, core.stream()))
    assert gbad.cpu().tolist() == [0.0, 1e5, -1e5, 1.0, -2.0] * 3
    # the data-parallel form: `/world` and nan_to_num(0, 1e5, -1e5) fused into the Adam pass (fuse_sanitize=1, gscale=1/world) on a
    # gradient that holds NaN / +-Inf, as the SUM all-reduce of two ranks can (training_loop.py:306-309 then :313)
    from oracle import losses_ref
    world = 2
    gsum = torch.randn(n) * 3
    gsum[5] = float('nan'); gsum[77] = float('inf'); gsum[1000] = -float('inf'); gsum[-1] = float('nan')
    g_ref = losses_ref.dp_postprocess(gsum.clone(), world)          # golden-pinned restatement of the reference's post-processing
    pr2 = p0.clone().requires_grad_(True); opt2 = torch.optim.Adam([pr2], lr=1e-3, betas=(0.0, 0.99), eps=1e-8)
    pr2.grad = g_ref.clone(); opt2.step()
    pg2 = p0.to(dev); m2 = torch.zeros(n, device=dev); v2 = torch.zeros(n, device=dev); gd = gsum.to(dev)
    core.check(core.lib().ldetr_adam_step_f32(core.ptr(pg2), core.ptr(gd), core.ptr(m2), core.ptr(v2), n, 1, 1e-3, 0.0, 0.99, 1e-8,
                                              1, 1.0 / world, 0.0, 1e5, -1e5, core.stream()))
    assert torch.isfinite(pg2).all() and torch.isfinite(v2).all()
    assert_close(pg2, pr2, 1e-6, 'adam fused sanitize')
    assert_close(m2, g_ref, 1e-6, 'first moment = sanitized gradient (beta1 = 0)')
    pe = torch.randn(n); pe_g = pe.to(dev)
    core.check(core.lib().ldetr_ema_lerp_f32(core.ptr(pe_g), core.ptr(pg), n, 0.9, core.stream()))
    assert_close(pe_g, pg.cpu().lerp(pe, 0.9), 1e-6, 'ema')
    # the G_ema lerp inside the optimiser pass (ldetr_adam_ema_step_f32) == Adam step followed by ldetr_ema_lerp_f32, bit for bit
    pa, ma, va = p0.to(dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    pb, mb, vb = p0.to(dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    ea, eb = pe.to(dev), pe.to(dev)
    gd2 = gsum.to(dev)
    core.check(core.lib().ldetr_adam_ema_step_f32(core.ptr(pa), core.ptr(gd2), core.ptr(ma), core.ptr(va), n, 1, 1e-3, 0.0, 0.99, 1e-8,
                                                  1, 0.5, 0.0, 1e5, -1e5, core.ptr(ea), 0.75, core.stream()))
    core.check(core.lib().ldetr_adam_step_f32(core.ptr(pb), core.ptr(gd2), core.ptr(mb), core.ptr(vb), n, 1, 1e-3, 0.0, 0.99, 1e-8,
                                              1, 0.5, 0.0, 1e5, -1e5, core.stream()))
    core.check(core.lib().ldetr_ema_lerp_f32(core.ptr(eb), core.ptr(pb), n, 0.75, core.stream()))
    assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb) and torch.equal(ea, eb)


# ------------------------------------------------------------------------------------------ LSAP (Hungarian)
def test_lsap_matches_scipy_bit_exact(dev):
    from scipy.optimize import linear_sum_assignment
    from layoutdetr_amd.hip import core
    rng = np.random.RandomState(0)
    for n in [1, 2, 3, 5, 9, 16, 17, 25, 50, 64]:      # n <= 16 and 17..64 are two instantiations (array / mask widths)
        batch = 64 if n <= 16 else 24
        cost = rng.rand(batch, n, n)
        cost[::3] = np.round(cost[::3] * 3) / 3  # heavy ties
        cost[1] = 0.5
        for maximize in (0, 1):
            c = torch.from_numpy(cost).to(dev)
            ri = torch.empty(batch, n, dtype=torch.int32, device=dev); ci = torch.empty_like(ri)
            core.check(core.lib().ldetr_lsap_f64(core.ptr(c), batch, n, maximize, core.ptr(ri), core.ptr(ci), core.stream()))
            for b in range(batch):
                r, cc = linear_sum_assignment(cost[b], maximize=bool(maximize))
                assert ri[b].cpu().tolist() == r.tolist() and ci[b].cpu().tolist() == cc.tolist()


# ------------------------------------------------------------------------------------------ detr_util/box_ops.py (north_star row ns-1)
def test_box_ops_bit_exact_vs_reference_golden_and_hungarian_indices_vs_scipy(dev):
    """box_cxcywh_to_xyxy / box_iou / generalized_box_iou on the GPU against vectors captured from the reference's own functions
    (tests/golden/box_ops.npz): every matrix BIT-exact in fp32 (the reference's operation order, no fma contraction, correctly rounded
    division); the Hungarian assignment on cost = -GIoU (one launch for the cost matrices of the whole batch + the device LSAP solve)
    index-exact against the fixture's scipy result and against live scipy on fresh layouts incl. duplicated boxes (ties)."""
    from scipy.optimize import linear_sum_assignment
    from layoutdetr_amd.detr_util import box_ops
    d = np.load(os.path.join(G, 'box_ops.npz'))
    for i in range(int(d['count'])):
        pred, tgt = torch.from_numpy(d[f'pred{i}']).to(dev), torch.from_numpy(d[f'tgt{i}']).to(dev)
        p_xyxy, t_xyxy = box_ops.box_cxcywh_to_xyxy(pred), box_ops.box_cxcywh_to_xyxy(tgt)
        assert np.array_equal(p_xyxy.cpu().numpy(), d[f'p_xyxy{i}']) and np.array_equal(box_ops.box_xyxy_to_cxcywh(p_xyxy).cpu().numpy(), d[f'back{i}'])
        iou, union = box_ops.box_iou(p_xyxy, t_xyxy)
        assert np.array_equal(iou.cpu().numpy(), d[f'iou{i}'], equal_nan=True), f'case {i}: iou not bit-exact'
        assert np.array_equal(union.cpu().numpy(), d[f'union{i}']), f'case {i}: union'
        assert np.array_equal(box_ops.generalized_box_iou(p_xyxy, t_xyxy).cpu().numpy(), d[f'giou{i}'], equal_nan=True), f'case {i}: giou not bit-exact'
        ri, ci, giou = box_ops.hungarian_match_giou(pred[None], tgt[None])
        assert np.array_equal(giou[0].cpu().numpy(), d[f'giou{i}'], equal_nan=True)
        assert ri[0].cpu().tolist() == d[f'row{i}'].tolist() and ci[0].cpu().tolist() == d[f'col{i}'].tolist(), f'case {i}: assignment'
    a, b = torch.from_numpy(d['rect_a']).to(dev), torch.from_numpy(d['rect_b']).to(dev)
    iou, union = box_ops.box_iou(a, b)
    assert np.array_equal(iou.cpu().numpy(), d['rect_iou']) and np.array_equal(union.cpu().numpy(), d['rect_union'])
    assert np.array_equal(box_ops.generalized_box_iou(a, b).cpu().numpy(), d['rect_giou'])
    with pytest.raises(AssertionError):
        box_ops.generalized_box_iou(torch.tensor([[0.5, 0.5, 0.4, 0.6]], device=dev), b)      # x1 < x0: rejected like the reference
    # a batch of layouts in one call against live scipy
    g = torch.Generator().manual_seed(5)
    for n in (9, 16, 33):
        B = 64
        pred = torch.cat([torch.rand(B, n, 2, generator=g) * 0.6 + 0.2, torch.rand(B, n, 2, generator=g) * 0.35 + 0.05], -1)
        tgt = torch.cat([torch.rand(B, n, 2, generator=g) * 0.6 + 0.2, torch.rand(B, n, 2, generator=g) * 0.35 + 0.05], -1)
        tgt[::4, 1] = tgt[::4, 0]; pred[1::4, 2] = pred[1::4, 0]; tgt[2::8] = pred[2::8]
        ri, ci, giou = box_ops.hungarian_match_giou(pred.to(dev), tgt.to(dev))
        cost = (-giou.cpu()).double().numpy()
        for bb in range(B):
            r, c = linear_sum_assignment(cost[bb])
            assert ri[bb].cpu().tolist() == r.tolist() and ci[bb].cpu().tolist() == c.tolist(), (n, bb)


# ------------------------------------------------------------------------------------------ layout metrics
def _metric_layouts():
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'metrics.npz'))
    L1 = [(d[f'l1_b{i}'], d[f'l1_l{i}']) for i in range(int(d['n1']))]
    L2 = [(d[f'l2_b{i}'], d[f'l2_l{i}']) for i in range(int(d['n2']))]
    return d, L1, L2


def test_layout_metrics_device_vs_reference_golden(dev):
    """Batched device scoring (pairwise IoU / DocSim + one device Hungarian solve per pair) against the values the reference's
    host functions produced, incl. a duplicated layout (tied assignments) and conditions present on one side only."""
    from layoutdetr_amd.metrics import metric_layoutnet as M
    d, L1, L2 = _metric_layouts()
    mi = np.asarray([M.compute_maximum_iou_for_layout(L1[i], L2[j]) for i, j in d['pairs']])
    md = np.asarray([M.compute_maximum_docsim_for_layout(L1[i], L2[j]) for i, j in d['pairs']])
    assert np.abs(mi - d['max_iou_pair']).max() <= 1e-6, np.abs(mi - d['max_iou_pair']).max()
    assert np.abs(md - d['max_docsim_pair']).max() <= 1e-6
    assert abs(M.compute_maximum_iou(L1, L2) - float(d['max_iou_corpus'])) <= 1e-6
    with pytest.raises(ValueError):
        M.compute_maximum_iou_for_layout(L1[0], (L2[0][0], L2[0][1] + 7))


def test_layout_metrics_device_vs_oracle_random(dev):
    """Larger random corpus: device pass vs oracle/metrics_ref.py (numpy + scipy)."""
    from layoutdetr_amd.metrics import metric_layoutnet as M
    from oracle import metrics_ref as R
    rng = np.random.RandomState(5)
    def layout(labels):
        n = len(labels)
        return (np.concatenate([rng.rand(n, 2) * 0.6 + 0.2, rng.rand(n, 2) * 0.35 + 0.05], -1).astype(np.float32), np.asarray(labels, dtype=np.int64))
    conds = [[0, 1, 1, 2, 2, 2, 3, 3, 3], [5, 5, 5], [0, 0, 0, 0, 0, 0], [1, 2]]
    L1 = [layout(list(rng.permutation(c))) for c in conds for _ in range(7)]
    L2 = [layout(list(rng.permutation(c))) for c in conds for _ in range(5)]
    assert abs(M.compute_maximum_iou(L1, L2) - R.compute_maximum_iou(L1, L2)) <= 1e-6
    b1 = torch.as_tensor(np.stack([l[0] for l in L1[:7]]), device=dev); l1 = torch.as_tensor(np.stack([l[1] for l in L1[:7]]), device=dev)
    b2 = torch.as_tensor(np.stack([l[0] for l in L2[:5]] + [L2[0][0], L2[1][0]]), device=dev)
    l2 = torch.as_tensor(np.stack([l[1] for l in L2[:5]] + [L2[0][1], L2[1][1]]), device=dev)
    got = M.maximum_scores_batched(b1, l1, b2, l2, 'docsim').cpu().numpy()
    L2x = L2[:5] + [L2[0], L2[1]]
    ref = np.asarray([R.compute_maximum_docsim_for_layout(L1[k], L2x[k]) for k in range(7)])
    assert np.abs(got - ref).max() <= 1e-6


def test_bottleneck_chain_fused_block_gradients(dev):
    """Three chained bottlenecks (stride-2 downsample block in the middle) with the trunk's gradient hand-offs switched on
    (block-input gradient = conv1 data gradient + identity/downsample gradient + previous block's ReLU mask in ONE kernel;
    no activation-gradient pass for conv1/conv2/chained block outputs) against plain torch autograd of the same network."""
    from layoutdetr_amd.training.detr_backbone import Bottleneck
    torch.manual_seed(71)
    blocks = [Bottleneck(32, 16, 1, downsample=True), Bottleneck(64, 32, 2, downsample=True), Bottleneck(128, 32)]
    for b in blocks:
        for n, p in b.named_parameters():
            p.data.normal_(0, 0.15)
        for m in b.modules():
            if hasattr(m, 'running_var'):
                m.weight.data.uniform_(0.6, 1.4); m.bias.data.normal_(0, 0.2); m.running_mean.normal_(0, 0.2); m.running_var.uniform_(0.5, 1.5)
    x = torch.randn(2, 32, 20, 20)
    g = torch.randn(2, 128, 10, 10)

    def ref_block(b, t):
        def cba(t, conv, bn, stride, pad, relu):
            w = conv.weight.detach().clone().requires_grad_(True); ws.append(w)
            sc = bn.weight * (bn.running_var + 1e-5).rsqrt(); sh = bn.bias - bn.running_mean * sc
            y = F.conv2d(t, w, stride=stride, padding=pad) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
            return F.relu(y) if relu else y
        o = cba(t, b.conv1, b.bn1, 1, 0, True)
        o = cba(o, b.conv2, b.bn2, b.conv2.stride, 1, True)
        idt = cba(t, b.downsample[0], b.downsample[1], b.downsample[0].stride, 0, False) if b.downsample is not None else t
        w = b.conv3.weight.detach().clone().requires_grad_(True); ws.append(w)
        sc = b.bn3.weight * (b.bn3.running_var + 1e-5).rsqrt(); sh = b.bn3.bias - b.bn3.running_mean * sc
        return F.relu(F.conv2d(o, w) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1) + idt)
    ws = []
    xr = x.clone().requires_grad_(True)
    t = F.relu(xr)            # the chain's input is a ReLU output, as inside the trunk
    for b in blocks:
        t = ref_block(b, t)
    ref_out = t.detach()
    t.backward(g)
    ref_w = [w.grad for w in ws]
    # HIP path with the chain flags the trunk sets
    for b in blocks:
        b.to(dev)
        for p in b.parameters():
            p.data = p.data.contiguous(memory_format=torch.channels_last) if p.ndim == 4 else p.data
            p.grad = None
    blocks[0].premask_out = True; blocks[1].mask_in = True; blocks[1].premask_out = True; blocks[2].mask_in = True
    xg = x.to(dev).requires_grad_(True)
    t = F.relu(xg).permute(0, 2, 3, 1).contiguous()
    for b in blocks:
        t = b(t)
    t.backward(g.permute(0, 2, 3, 1).contiguous().to(dev))
    assert_close(t.permute(0, 3, 1, 2), ref_out, 2e-5, 'forward')
    assert_close(xg.grad, xr.grad, 2e-5, 'dx')
    got_w = []
    for b in blocks:
        got_w += [b.conv1.weight.grad, b.conv2.weight.grad] + ([b.downsample[0].weight.grad] if b.downsample is not None else []) + [b.conv3.weight.grad]
    for i, (a, r) in enumerate(zip(got_w, ref_w)):
        assert_close(a, r, 3e-5, f'dw[{i}]')


@pytest.mark.parametrize('B,H,L,dh,causal', [(2, 4, 40, 192, True), (3, 3, 21, 64, True), (2, 2, 70, 96, False), (2, 4, 40, 192, False), (2, 8, 19, 32, True), (2, 8, 64, 32, True), (3, 4, 37, 32, True),
                                             (2, 2, 256, 192, False), (2, 2, 200, 64, True), (1, 3, 130, 160, False), (2, 1, 64, 128, True),
                                             (2, 2, 300, 64, True), (1, 2, 520, 96, False)])
def test_attention_wide_heads_and_causal(dev, B, H, L, dh, causal):
    """Packed self-attention for 32..192-wide heads, forward + backward, with the decoder's causal mask on top of a ragged
    key-padding mask (BertSelfAttention shapes of the text encoder / LM decoder)."""
    from layoutdetr_amd.hip.attention import _AttnPackedFn
    torch.manual_seed(21)
    d = H * dh
    qkv = torch.randn(B * L, 3 * d) * 0.5
    kpm = torch.zeros(B, L, dtype=torch.bool); kpm[0, L - 5:] = True; kpm[-1, L // 2:] = True
    g = torch.randn(B * L, d)
    qr = qkv.clone().requires_grad_(True)
    q, k, v = [t.reshape(B, L, H, dh).permute(0, 2, 1, 3) for t in qr.split(d, dim=1)]
    s = (q @ k.transpose(-1, -2)) / math.sqrt(dh)
    s = s.masked_fill(kpm[:, None, None, :], float('-inf'))
    if causal:
        s = s.masked_fill(torch.triu(torch.ones(L, L, dtype=torch.bool), 1), float('-inf'))
    ref = (torch.softmax(s, -1) @ v).permute(0, 2, 1, 3).reshape(B * L, d)
    ref.backward(g)
    qg = qkv.to(dev).requires_grad_(True)
    out = _AttnPackedFn.apply(qg, None, kpm.to(torch.uint8).to(dev), B, H, L, 0.0, causal)
    out.backward(g.to(dev))
    assert_close(out, ref.detach(), 1e-5, 'out')
    assert_close(qg.grad, qr.grad, 2e-5, 'dqkv')


@pytest.mark.parametrize('rows,V,eps', [(37, 30524, 0.1), (64, 1000, 0.0), (5, 70, 0.1), (300, 4098, 0.2)])
def test_softmax_cross_entropy_label_smoothing(dev, rows, V, eps):
    """Fused label-smoothed softmax cross entropy (LM decoder loss) vs F.cross_entropy: value and gradient, ignored rows,
    a vocabulary that is not a multiple of 4, an upstream gradient != 1, and the all-ignored corner."""
    from layoutdetr_amd.training.med import softmax_cross_entropy
    torch.manual_seed(31)
    x = torch.randn(rows, V) * 3; t = torch.randint(0, V, (rows,)); t[::4] = -100
    xr = x.clone().requires_grad_(True)
    ref = F.cross_entropy(xr, t, ignore_index=-100, label_smoothing=eps); (ref * 2.5).backward()
    if V % 4:
        xg = torch.zeros(rows, (V + 3) // 4 * 4, device=dev)[:, :V]; xg.copy_(x.to(dev)); xg.requires_grad_(True)   # 16-byte aligned row pitch
    else:
        xg = x.to(dev).requires_grad_(True)
    out = softmax_cross_entropy(xg, t.to(dev), -100, eps); (out * 2.5).backward()
    assert abs(out.item() - ref.item()) <= 2e-6 * abs(ref.item()) + 1e-6
    assert_close(xg.grad, xr.grad, 2e-5, 'dlogits')
    assert float(xg.grad[::4].abs().max()) == 0.0


@pytest.mark.parametrize('B,T,V,d', [(144, 40, 30524, 768), (3, 7, 50, 64), (2, 512, 1000, 128)])
def test_token_embedding_gather_and_scatter(dev, B, T, V, d):
    """word + position embedding of the text models vs nn.Embedding: the gather is bit-exact, the atomic scatter of the backward
    matches aten's embedding_dense_backward (padding_idx rows get no gradient); 144 x 40 tokens is the B=16 hot-path shape, the
    size class where aten switches to its sort-based backward."""
    import torch.nn as nn
    from layoutdetr_amd.training.med import _TokenEmbeddingFn
    torch.manual_seed(5)
    word = nn.Embedding(V, d, padding_idx=0); pos = nn.Embedding(max(T, 512), d)
    ids = torch.randint(0, V, (B, T)); ids[:, -3:] = 0; ids[0, 0] = V - 1
    ref = word(ids) + pos(torch.arange(T)[None])
    g = torch.randn(B, T, d)
    ref.backward(g)
    wg = word.weight.detach().clone().to(dev).requires_grad_(True); pg = pos.weight.detach().clone().to(dev).requires_grad_(True)
    out = _TokenEmbeddingFn.apply(ids.to(dev), wg, pg, 0)
    assert torch.equal(out.cpu(), ref.detach().reshape(B * T, d))
    out.backward(g.reshape(B * T, d).to(dev))
    assert_close(wg.grad, word.weight.grad, 2e-5, 'dword')
    assert_close(pg.grad, pos.weight.grad, 2e-5, 'dpos')
    assert float(wg.grad[0].abs().max()) == 0.0


def test_background_resize_normalize_vs_pillow_golden(dev):
    """Device resize + normalise of the dataset item's page background against Pillow's own outputs and the
    reference's fp32 normalisation lines (tests/golden/resample.npz): uint8 and float results bit-exact."""
    from layoutdetr_amd.training.dataset_layoutganpp import background_to_tensor
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'resample.npz'))
    for i in range(int(d['n'])):
        img = torch.from_numpy(d[f'in{i}']).to(dev)
        s = d[f'u8_{i}'].shape[0]
        out, u8 = background_to_tensor(img, s, return_u8=True)
        assert torch.equal(u8.cpu(), torch.from_numpy(d[f'u8_{i}'])), f'case {i}: uint8'
        assert torch.equal(out.cpu(), torch.from_numpy(d[f'out{i}'])), f'case {i}: float CHW'


@pytest.mark.parametrize('n,H,W,S', [(2, 1024, 1024, 256), (3, 700, 1000, 256), (1, 300, 200, 512), (2, 4000, 90, 64), (1, 64, 20000, 128)])
def test_background_resize_normalize_vs_oracle_full_size(dev, n, H, W, S):
    """Same at the dataset's real page size (1024 x 1024 -> 256) and ragged / extreme shapes (up-scaling, a row segment too wide
    for the LDS staging) against the CPU oracle, bit-exact; plus batch independence."""
    from layoutdetr_amd.training.dataset_layoutganpp import background_to_tensor
    from oracle import resample_ref
    rng = np.random.default_rng(n * 1000 + S)
    imgs = rng.integers(0, 256, (n, H, W, 3), dtype=np.uint8)
    imgs[0, : H // 2] = (np.arange(W)[None, :, None] * 255 // max(W - 1, 1)).astype(np.uint8)      # smooth half
    out, u8 = background_to_tensor(torch.from_numpy(imgs).to(dev), S, return_u8=True)
    for i in range(n):
        ref_u8 = resample_ref.resize_antialias_u8(imgs[i], S, S)
        assert np.array_equal(u8[i].cpu().numpy(), ref_u8), f'image {i}: uint8'
        assert np.array_equal(out[i].cpu().numpy(), resample_ref.normalize_chw(ref_u8)), f'image {i}: float'
    one = background_to_tensor(torch.from_numpy(imgs[n - 1]).to(dev), S)
    assert torch.equal(one, out[n - 1])


@pytest.mark.parametrize('B,N,seed', [(16, 9, 0), (3, 10, 1), (5, 16, 2), (2, 1, 3), (4, 2, 4), (3, 25, 5), (2, 50, 6), (2, 64, 7)])
def test_fused_layout_losses_match_reference_formulation(dev, B, N, seed):
    """csrc/layout_loss.hip: mse / gIoU / overlap / alignment of generated boxes and their gradients in one
    launch, against the oracle's restatement of the reference functions (golden-pinned in tests/test_oracle_golden.py) run through
    autograd on the CPU: ragged masks, padded slots with arbitrary contents (compute_alignment looks at them), touching /
    nested / disjoint boxes, upstream weights per term."""
    from layoutdetr_amd.metrics.metric_layoutnet import layout_losses_fused
    from oracle import losses_ref
    g = torch.Generator().manual_seed(100 + seed)
    xy = torch.rand(B, N, 2, generator=g) * 0.6 + 0.2; wh = torch.rand(B, N, 2, generator=g) * 0.35 + 0.05
    fake = torch.cat([xy, wh], -1)
    real = torch.cat([torch.rand(B, N, 2, generator=g) * 0.6 + 0.2, torch.rand(B, N, 2, generator=g) * 0.35 + 0.05], -1)
    if N >= 3:
        fake[0, 1] = fake[0, 0] * torch.tensor([1.0, 1.0, 0.5, 0.5])        # nested box, shared centre
        fake[0, 2, :2] = fake[0, 0, :2] + 0.9                                 # far away: disjoint
    valid = torch.ones(B, N, dtype=torch.bool)
    if N >= 2:
        valid[-1, N // 2:] = False
        if B > 1:
            valid[1, -1] = False
    wts = torch.tensor([100.0, 4.0, 7.0, 17.0])
    fr = fake.clone().requires_grad_(True)
    ref_terms = [F.mse_loss(fr[valid], real[valid]), losses_ref.generalized_iou_loss(fr[valid], real[valid]),
                 losses_ref.compute_overlap(fr, valid), losses_ref.compute_alignment(fr, valid)]
    up = torch.rand(B, generator=g) + 0.5
    total_ref = ref_terms[0] * wts[0] + ref_terms[1] * wts[1] + (ref_terms[2] * up).sum() * wts[2] + (ref_terms[3] * up).mean() * wts[3]
    total_ref.backward()
    fg = fake.to(dev).requires_grad_(True)
    out = layout_losses_fused(fg, real.to(dev), valid.to(dev))
    total = out[0] * wts[0] + out[1] * wts[1] + (out[2] * up.to(dev)).sum() * wts[2] + (out[3] * up.to(dev)).mean() * wts[3]
    total.backward()
    for name, a, b in zip(('mse', 'giou', 'overlap', 'alignment'), out, ref_terms):
        assert_close(a, b.detach(), 2e-5, name)
    assert_close(fg.grad, fr.grad, 5e-5, 'd bbox')


@pytest.mark.parametrize('B,O,I,K,cl', [(16, 512, 512, 3, True), (4, 64, 128, 3, False), (2, 32, 32, 1, False), (3, 100, 36, 3, True)])
def test_demod_coefficients_fwd_bwd(dev, B, O, I, K, cl):
    """Fused demodulation coefficients (csrc/demod.hip) vs the reference's formulation through autograd
    (networks_stylegan2.py:57-61: (w[None] * s[:, None, :, None, None]).square().sum([2, 3, 4]) + 1e-8).rsqrt()), both parameter
    memory layouts, gradients to the weight and to the styles."""
    from layoutdetr_amd.hip.modconv import _DemodFn
    torch.manual_seed(60)
    w = torch.randn(O, I, K, K) * 0.3; s = torch.randn(B, I); g = torch.randn(B, O)
    wr = w.clone().requires_grad_(True); sr = s.clone().requires_grad_(True)
    ref = ((wr.unsqueeze(0) * sr.reshape(B, 1, I, 1, 1)).square().sum(dim=[2, 3, 4]) + 1e-8).rsqrt()
    ref.backward(g)
    wd = w.to(dev)
    if cl:
        wd = wd.contiguous(memory_format=torch.channels_last)
    wd.requires_grad_(True); sd = s.to(dev).requires_grad_(True)
    out = _DemodFn.apply(wd, sd)
    out.backward(g.to(dev))
    assert_close(out, ref.detach(), 5e-6, 'dcoefs')
    assert_close(wd.grad, wr.grad, 2e-5, 'dweight')
    assert_close(sd.grad, sr.grad, 2e-5, 'dstyles')


@pytest.mark.parametrize('B,R,Ci,Co,up', [(2, 16, 64, 64, 1), (2, 8, 128, 64, 2), (3, 16, 64, 128, 2), (2, 32, 96, 64, 1), (2, 4, 512, 512, 2),
                                          (4, 256, 32, 32, 1)])   # the 256x256 32-channel layer: register-resident filter bank (csrc/conv_c32.hip: fwd, dX) + operand-streaming weight gradient
def test_modulated_conv_layers_vs_oracle(dev, B, R, Ci, Co, up):
    """One StyleGAN2 synthesis layer (modulate -> 3x3 conv or transposed conv + 4x4 FIR -> demodulate -> bias -> lrelu * sqrt 2,
    networks_stylegan2.py:30-75,307-326) against the oracle's non-fused formulation, forward and all gradients, at channel counts
    that take the scalar-addressed (FAST) kernels for the conv, the transposed conv, their data and weight gradients."""
    from layoutdetr_amd.hip import modconv
    torch.manual_seed(70 + R + up)
    x = torch.randn(B, Ci, R, R); w = torch.randn(Co, Ci, 3, 3) / math.sqrt(Ci * 9); s = torch.randn(B, Ci) * 0.5 + 1.0; b = torch.randn(Co) * 0.1
    if B * R * R * Co > 2_000_000:
        # among millions of lrelu outputs a few pre-activations lie within rounding distance of 0 and take the other slope in one of two
        # fp32 evaluations (seen: 3e-5 of dx off, dw off by 4e-3): a bias of +8 sigma keeps this large case on one branch
        b = b + 8.0
    f = ops_ref.setup_filter([1, 3, 3, 1])
    xr, wr, sr, br = [t.clone().requires_grad_(True) for t in (x, w, s, b)]
    y = ops_ref.modulated_conv2d(xr, wr, sr, up=up, padding=1, resample_filter=f, demodulate=True, flip_weight=(up == 1))
    y = ops_ref.bias_act(y, br, act='lrelu', gain=math.sqrt(2))
    g = torch.randn_like(y); y.backward(g)
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev).requires_grad_(True)
    wd = w.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    sd = s.to(dev).requires_grad_(True); bd = b.to(dev).requires_grad_(True)
    out = modconv.modconv3x3(xd, wd, sd, bd) if up == 1 else modconv.modconv3x3_up2(xd, wd, sd, bd, f.to(dev))
    out.backward(g.permute(0, 2, 3, 1).contiguous().to(dev))
    assert_close(out.permute(0, 3, 1, 2), y.detach(), 2e-5, 'y')
    assert_close(xd.grad.permute(0, 3, 1, 2), xr.grad, 5e-5, 'dx')
    assert_close(wd.grad, wr.grad, 5e-5, 'dw')
    assert_close(sd.grad, sr.grad, 1e-4, 'dstyles')
    assert_close(bd.grad, br.grad, 5e-5, 'dbias')


@pytest.mark.parametrize('tokens,N,K', [(144, 256, 256), (160, 2048, 256), (1024, 256, 256), (144, 256, 2048), (1024, 512, 512), (9000, 256, 256), (70, 36, 52)])
def test_gemm_pair_data_and_weight_gradient(dev, tokens, N, K):
    """ldetr_gemm_pair_f32: the data gradient dX = dY W (+ residual) and the weight gradient dW += dY^T X (+ bias row sums) of one
    linear layer through one call -- one launch when both are small-tile problems, two otherwise (long K, many tokens, ragged
    sizes): results must equal the separate calls and the fp64 reference."""
    from layoutdetr_amd.hip import core
    torch.manual_seed(90)
    dy = torch.randn(tokens, N); w = torch.randn(N, K) / math.sqrt(K); x = torch.randn(tokens, K); res = torch.randn(tokens, K)
    gw0 = torch.randn(N, K); gb0 = torch.randn(N)
    dyd, wd, xd, resd = [t.to(dev) for t in (dy, w, x, res)]
    use_rs = N % 4 == 0
    gw = gw0.to(dev).clone(); gb = gb0.to(dev).clone(); dx = torch.empty(tokens, K, device=dev)
    core.gemm_pair(dict(A=dyd, B=wd, ta=0, tb=1, M=tokens, N=K, K=N, out=dx, ep=core.epilogue(alpha=0.5, residual=resd)),
                   dict(A=dyd, B=xd, ta=1, tb=1, M=N, N=K, K=tokens, out=gw, ep=core.epilogue(alpha=0.5, accumulate=True, a_rowsum=gb if use_rs else None)))
    assert_close(dx, (0.5 * (dy.double() @ w.double()) + res.double()).float(), 3e-6, 'dX')
    assert_close(gw, (gw0.double() + 0.5 * (dy.double().t() @ x.double())).float(), 3e-6, 'dW')
    if use_rs:
        assert_close(gb, (gb0.double() + dy.double().sum(0)).float(), 3e-6, 'dbias')
    dx2 = core.gemm(dyd, wd, 0, 1, tokens, K, N, ep=core.epilogue(alpha=0.5, residual=resd))
    assert_close(dx, dx2, 1e-6, 'pair vs single dX')


def test_fast_and_generic_instantiations_agree(dev, tmp_path):
    """The scalar-addressed (FAST) kernels and the paired launch only change how addresses are formed and how work is batched:
    a subprocess with them switched off (LDETR_DEBUG="FAST_LOADS=0,SMALL_FAST=0,GEMM_PAIR=0,SPLIT_BF16=0") must produce the same
    conv / linear forward, data gradient and weight gradient as this process, to the last bit for the tiled conv kernels
    (same reduction order) and to 1e-6 where split-K decisions may differ."""
    import subprocess, sys
    script = r'''
import sys, math, numpy as np, torch
import torch.nn.functional as F
from layoutdetr_amd.hip import conv
from layoutdetr_amd.hip.linear import linear
dev = torch.device('cuda:0'); out = {}
torch.manual_seed(123)
for i, (N, H, Ci, Co, k, s, p) in enumerate([(2, 32, 64, 128, 3, 1, 1), (2, 32, 128, 64, 3, 2, 1), (2, 16, 256, 256, 1, 1, 0), (4, 32, 64, 256, 1, 1, 0), (4, 32, 256, 128, 1, 1, 0)]):
    x = torch.randn(N, H, H, Ci, device=dev, requires_grad=True); w = (torch.randn(Co, Ci, k, k, device=dev) / math.sqrt(Ci * k * k)).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = conv.conv2d_nhwc(x, w, None, None, None, s, p, relu=False)
    g = torch.randn_like(y); y.backward(g)
    out[f'y{i}'] = y.detach().cpu().numpy(); out[f'dx{i}'] = x.grad.cpu().numpy(); out[f'dw{i}'] = w.grad.cpu().numpy()
xl = torch.randn(1024, 256, device=dev, requires_grad=True); wl = (torch.randn(2048, 256, device=dev) / 16).requires_grad_(True); bl = torch.randn(2048, device=dev, requires_grad=True)
yl = linear(xl, wl, bl); yl.backward(torch.randn_like(yl))
out['yl'] = yl.detach().cpu().numpy(); out['dxl'] = xl.grad.cpu().numpy(); out['dwl'] = wl.grad.cpu().numpy(); out['dbl'] = bl.grad.cpu().numpy()
np.savez(sys.argv[1], **out)
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    # SPLIT_BF16=0 on both sides: the bf16 split path exists for FAST operands only -> compare the f32 MFMA pipe with itself
    for tag, env in (('fast', {'LDETR_DEBUG': 'SPLIT_BF16=0'}), ('generic', {'LDETR_DEBUG': 'FAST_LOADS=0,SMALL_FAST=0,GEMM_PAIR=0,SPLIT_BF16=0'})):
        path = str(tmp_path / f'{tag}.npz')
        e = dict(os.environ); e.update(env); e['PYTHONPATH'] = root + os.pathsep + e.get('PYTHONPATH', '')
        subprocess.run([sys.executable, '-c', script, path], check=True, env=e, cwd=root, timeout=300, stdin=subprocess.DEVNULL)
        res[tag] = np.load(path)
    for key in res['fast'].files:
        a, b = res['fast'][key], res['generic'][key]
        if key.startswith(('y', 'dx')) and not key.endswith('l'):
            assert np.array_equal(a, b), f'{key}: FAST and generic tiled kernels differ'
        else:
            err = np.abs(a.astype(np.float64) - b).max() / (np.abs(b).max() + 1e-12)
            assert err <= 1e-6, f'{key}: rel err {err:.2e}'


# ------------------------------------------------------------------------------------------ bf16 split pipe vs f32 MFMA pipe
def _both_pipes(fn):
    """fn() on the f32 MFMA pipe and on the bf16 pipe with the exact three-way operand split (ldetr_set_split_bf16)."""
    from layoutdetr_amd.hip import core
    L = core.lib()
    prev = L.ldetr_set_split_bf16(0)
    try:
        f32 = fn()
        L.ldetr_set_split_bf16(15)
        sp = fn()
    finally:
        L.ldetr_set_split_bf16(prev)
    return f32, sp


def _err(x, ref):
    d = (x.double().cpu() - ref).abs()
    return d.max().item() / ref.abs().max().item(), (d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()


@pytest.mark.parametrize('ta,tb,M,N,K', [(0, 0, 4096, 512, 1152), (0, 1, 16384, 256, 512), (1, 1, 1024, 1024, 8192), (0, 0, 4096, 2048, 256)])
def test_split_bf16_gemm_is_fp32_equivalent(dev, ta, tb, M, N, K):
    """The split path (6 bf16 MFMAs per k16 on hi/mid/lo parts of the fp32 operands) against an fp64 contraction: at least as
    accurate as the f32 MFMA path on operands with mixed exponents, for every operand view (k-contiguous / row-contiguous)."""
    from layoutdetr_amd.hip import core
    torch.manual_seed(40)
    A = torch.randn((K, M) if ta else (M, K)) * torch.exp(torch.randn(K, 1) if ta else torch.randn(1, K))
    B = torch.randn((K, N) if tb else (N, K)) * torch.exp(0.5 * (torch.randn(K, 1) if tb else torch.randn(1, K)))
    ref = (A.double().t() if ta else A.double()) @ (B.double() if tb else B.double().t())
    Ad, Bd = A.to(dev), B.to(dev)
    f32, sp = _both_pipes(lambda: core.gemm(Ad, Bd, ta, tb, M, N, K).clone())
    (m0, r0), (m1, r1) = _err(f32, ref), _err(sp, ref)
    print(f'gemm ta={ta} tb={tb} {M}x{N}x{K}: f32 pipe max {m0:.2e} rms {r0:.2e} | bf16 split max {m1:.2e} rms {r1:.2e}')
    assert not torch.equal(f32, sp), 'both runs took the same path: the split tiles were not exercised'
    assert r1 <= 1.25 * r0 + 1e-9 and m1 <= 2.0 * m0 + 1e-9 and m1 < 5e-6


def test_conv_c32_split_pipe_is_fp32_equivalent_and_non_finite_safe(dev):
    """csrc/conv_c32.hip (3x3, 32 -> 32 channels, 256x256: forward with per-sample input-channel factors + per-sample output factors + bias +
    lrelu, and the data gradient = the transposed, tap-flipped bank) on the f32 MFMA pipe and on the bf16 pipe with the exact operand split:
    both against fp64 (the split at least as close), and with +-Inf / NaN / FLT_MAX planted in the activations the split kernel must return
    exactly the f32 kernel's non-finite pattern (a segment that meets one recomputes itself on the f32 pipe) and untouched values elsewhere."""
    import ctypes
    from layoutdetr_amd.hip import core
    torch.manual_seed(43)
    N, H, C = 4, 256, 32
    x = torch.randn(N, H, H, C) * torch.exp(0.7 * torch.randn(1, 1, 1, C)); w = torch.randn(C, 3, 3, C) * 0.1          # NHWC, OHWI
    st = torch.rand(N, C) + 0.5; dm = torch.rand(N, C) + 0.5; b = torch.randn(C) * 0.1
    L = core.lib()

    def run(xin, transposed):
        xd, wd, sd, dd, bd = [t.to(dev).contiguous() for t in (xin, w, st, dm, b)]
        y = torch.empty(N, H, H, C, device=dev)
        xt = core.tensor4_nhwc(xd)
        if not transposed:
            ep = core.epilogue(col_bias=bd, samp_scale=dd, act=core.ACT_LRELU, act_alpha=0.2, act_gain=math.sqrt(2))
            core.check(L.ldetr_conv2d_fwd_f32(core.ptr(xd), ctypes.byref(xt), core.ptr(wd), C, 3, 3, 1, 1, core.ptr(y), C, H, H, core.ptr(sd), C,
                                              ctypes.byref(ep), core.stream()), 'conv fwd')
        else:
            core.check(L.ldetr_conv2d_bwd_data_f32(core.ptr(xd), ctypes.byref(xt), core.ptr(wd), C, 3, 3, 1, 1, core.ptr(y), C, H, H, core.ptr(dd), C,
                                                   None, core.stream()), 'conv bwd data')
        torch.cuda.synchronize()
        return y.cpu()
    for transposed in (False, True):
        xr = x.double().permute(0, 3, 1, 2)
        if not transposed:
            ref = F.conv2d(xr * st.double()[:, :, None, None], w.double().permute(0, 3, 1, 2), padding=1) * dm.double()[:, :, None, None] + b.double()[None, :, None, None]
            ref = F.leaky_relu(ref, 0.2) * math.sqrt(2)
        else:
            ref = F.conv_transpose2d(xr * dm.double()[:, :, None, None], w.double().permute(0, 3, 1, 2), padding=1)
        ref = ref.permute(0, 2, 3, 1)
        f32, sp = _both_pipes(lambda: run(x, transposed))
        (m0, r0), (m1, r1) = _err(f32, ref), _err(sp, ref)
        print(f'conv_c32 transposed={transposed}: f32 pipe max {m0:.2e} rms {r0:.2e} | bf16 split max {m1:.2e} rms {r1:.2e}')
        assert not torch.equal(f32, sp), 'both runs took the same kernel'
        assert r1 <= 1.25 * r0 + 1e-9 and m1 <= 2.0 * m0 + 1e-9 and m1 < 5e-6
        xp = x.clone()
        xp[0, 10, 40, 3] = float('inf'); xp[1, 200, 77, 9] = float('-inf'); xp[2, 0, 0, 31] = float('nan'); xp[3, 255, 255, 0] = 3.4e38; xp[3, 100, 31, 5] = float('inf')
        f32p, spp = _both_pipes(lambda: run(xp, transposed))
        bad = ~torch.isfinite(f32p)
        assert bad.any() and torch.equal(bad, ~torch.isfinite(spp)), 'non-finite pattern differs between the pipes'
        assert torch.equal(torch.isnan(f32p), torch.isnan(spp)) and torch.equal(f32p[bad & ~torch.isnan(f32p)], spp[bad & ~torch.isnan(spp)]), 'Inf / NaN classes differ'
        touched = bad | (f32p != f32)                                   # outputs whose 3x3 window holds a planted value
        seg = touched.reshape(N, H, H // 32, 32 * C).any(-1)             # their 32-pixel segments (the unit the f32 redo works on)
        clean = ~seg[..., None].expand(N, H, H // 32, 32 * C).reshape(N, H, H, C)
        assert torch.equal(spp[clean], sp[clean]), 'a segment without planted values changed'
        fin = ~bad & ~clean
        assert ((spp[fin].double() - f32p[fin].double()).abs() <= 2e-6 * f32p[fin].double().abs() + 1e-5).all(), 'finite values of a recomputed segment'


def test_split_bf16_conv_fwd_bwd_is_fp32_equivalent(dev):
    """3x3 conv forward, data gradient and weight gradient (tap-addressed, transposed-tap and pixel-major operand views) on both
    pipes against fp64."""
    from layoutdetr_amd.hip import conv
    torch.manual_seed(41)
    N, H, C, O = 16, 64, 128, 128
    x = torch.randn(N, C, H, H) * torch.exp(torch.randn(1, C, 1, 1)); w = torch.randn(O, C, 3, 3) * (0.05 * torch.exp(0.5 * torch.randn(O, 1, 1, 1)))
    g = torch.randn(N, O, H, H)
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    yr = F.conv2d(xr, wr, padding=1); yr.backward(g.double())

    def run():
        xg = x.permute(0, 2, 3, 1).contiguous().to(dev).requires_grad_(True)
        wg = w.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        y = conv.conv2d_nhwc(xg, wg, None, None, None, stride=1, pad=1, relu=False)
        y.backward(g.permute(0, 2, 3, 1).contiguous().to(dev))
        return y.detach().permute(0, 3, 1, 2), xg.grad.permute(0, 3, 1, 2), wg.grad.clone()
    f32, sp = _both_pipes(run)
    for name, a, b, ref in zip(('y', 'dx', 'dw'), f32, sp, (yr.detach(), xr.grad, wr.grad)):
        (m0, r0), (m1, r1) = _err(a, ref), _err(b, ref)
        print(f'conv {name}: f32 pipe max {m0:.2e} rms {r0:.2e} | bf16 split max {m1:.2e} rms {r1:.2e}')
        assert name == 'dw' or not torch.equal(a, b), name + ': the split tiles were not exercised'
        assert r1 <= 1.25 * r0 + 1e-9 and m1 <= 2.0 * m0 + 1e-9 and m1 < 5e-6, name


@pytest.mark.gpu
@pytest.mark.parametrize('B,P,C', [(16, 4096, 32), (2, 65536, 32), (16, 64, 512), (1, 3000, 64), (3, 50, 1280), (16, 16, 512)])
def test_row_reductions_all_paths_vs_fp64(dev, B, P, C):
    """ldetr_colsum_f32 / ldetr_act_bwd_reduce_f32 / ldetr_mul_reduce_f32 (training/networks_stylegan2.py:66-72, bias_act backward + bias / demodulation
    gradients) over shapes that take the direct-atomic path (<= 32 sharing blocks) and the two-stage path (partial sums through the workspace + rr_finish_kernel)
    with per-sample outputs and with an output shared by the batch, more than 1024 channels (channel chunks in grid.z) and ragged row counts; outputs
    accumulate INTO their destination.  Run twice (the workspace ring moves on)."""
    from layoutdetr_amd.hip import core
    from layoutdetr_amd.hip.linear import act_backward
    from layoutdetr_amd.hip.modconv import _mul_reduce
    g = torch.Generator(device='cpu').manual_seed(B * 1000 + C)
    dy = torch.randn(B * P, C, generator=g); y = torch.randn(B * P, C, generator=g)
    bias = torch.randn(C, generator=g); demod = torch.rand(B, C, generator=g) + 0.5; scale = torch.randn(B, C, generator=g)
    dyd, yd, bd, dd, sd = (t.to(dev) for t in (dy, y, bias, demod, scale))
    alpha, gain = 0.2, 2 ** 0.5
    for rep in range(2):
        # column sums per sample
        red = core.colsum(dyd, B)
        ref = dy.double().reshape(B, P, C).sum(1)
        assert rel_err(red, ref) < 2e-6, ('colsum', rep)
        # lrelu backward + bias gradient (shared by the batch, accumulated into an existing buffer) + demodulation gradient (per sample)
        dbias = torch.full((C,), 3.0, device=dev)
        dv, _, ddemod = act_backward(dyd, yd, core.ACT_LRELU, alpha, gain, True, bias=bd, demod=dd, want_ddemod=True, B=B, dbias_out=dbias)
        dvr = dy.double() * torch.where(y > 0, gain, gain * alpha)
        pre = torch.where(y > 0, y.double() / gain, y.double() / (gain * alpha)) - bias.double()
        ddr = (dvr * pre).reshape(B, P, C).sum(1) / demod.double()
        assert rel_err(dv, dvr) < 1e-6 and rel_err(dbias - 3.0, dvr.sum(0)) < 5e-6 and rel_err(ddemod, ddr) < 5e-6, ('act_bwd_reduce', rep)
        # modulation backward: out = a * scale[b], red[b] = sum_rows a * x
        out, red2 = _mul_reduce(dyd, yd, sd, B, P, C)
        assert rel_err(out.reshape(B, P, C), dy.double().reshape(B, P, C) * scale.double()[:, None]) < 1e-6, ('mul_reduce out', rep)
        assert rel_err(red2, (dy.double() * y.double()).reshape(B, P, C).sum(1)) < 5e-6, ('mul_reduce red', rep)
