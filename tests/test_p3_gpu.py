"""GPU parity of the plane-format ("P3") convolution engine (csrc/p3_engine.hip) through the C ABI.

The reference runs the ResNet-50 trunk through ATen's fp32 conv2d and its autograd (training/detr_backbone.py:98-114 via torchvision's
resnet50); the yardstick here is torch's conv2d evaluated in float64 on the host, and the bar is fp32-equivalence: the bf16-pipe result
must be as close to the float64 value as an fp32 evaluation of the same contraction is (tolerances written per test).  The P3 <-> fp32
conversions are exact, bit for bit, for every fp32 value including +-Inf, NaN and values next to FLT_MAX; non-finite operands must come
out in the same class (NaN / +Inf / -Inf) as an fp32 convolution produces, because training_loop.py:306-309's
nan_to_num(nan=0, posinf=1e5, neginf=-1e5) maps the classes to different gradients."""
import ctypes
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def L(dev):
    from layoutdetr_amd.hip import core
    return core.lib()


def _core():
    from layoutdetr_amd.hip import core
    return core


def p3_split(L, x2d):
    core = _core()
    rows, C = x2d.shape
    out = torch.empty(rows * C * 6, dtype=torch.uint8, device=x2d.device)
    core.check(L.ldetr_p3_split_f32(core.ptr(x2d), x2d.stride(0), core.ptr(out), rows, C, core.stream()), 'split')
    return out


def p3_merge(L, p, rows, C):
    core = _core()
    out = torch.empty(rows, C, dtype=torch.float32, device=p.device)
    core.check(L.ldetr_p3_merge_f32(core.ptr(p), core.ptr(out), C, rows, C, core.stream()), 'merge')
    return out


def p3_weight_bwd(L, w, scale=None):
    core = _core()
    Co, kh, kw, Ci = w.shape
    out = torch.empty(Ci * kh * kw * Co * 6, dtype=torch.uint8, device=w.device)
    core.check(L.ldetr_p3_weight_bwd(core.ptr(w), core.ptr(scale), core.ptr(out), Co, kh, kw, Ci, core.stream()), 'weight_bwd')
    return out


def epilogue(scale=None, shift=None, residual_p3=None, residual_f32=None, mask_p3=None, relu=False):
    from layoutdetr_amd import _lib
    ep = _lib.P3Epilogue()
    ep.alpha = 1.0
    ep.col_scale = scale.data_ptr() if scale is not None else None
    ep.col_bias = shift.data_ptr() if shift is not None else None
    ep.residual_p3 = residual_p3.data_ptr() if residual_p3 is not None else None
    ep.residual_f32 = residual_f32.data_ptr() if residual_f32 is not None else None
    ep.relu_mask_p3 = mask_p3.data_ptr() if mask_p3 is not None else None
    ep.relu = 1 if relu else 0
    return ep


def conv_ref64(x, w, stride, pad):
    """x [N,H,W,Ci], w [Co,KH,KW,Ci] -> [N,OH,OW,Co] in float64 on the host."""
    return F.conv2d(x.double().cpu().permute(0, 3, 1, 2), w.double().cpu().permute(0, 3, 1, 2), stride=stride, padding=pad).permute(0, 2, 3, 1)


def conv_f32_host(x, w, stride, pad):
    return F.conv2d(x.cpu().permute(0, 3, 1, 2), w.cpu().permute(0, 3, 1, 2), stride=stride, padding=pad).permute(0, 2, 3, 1)


def err(a, ref):
    a, ref = a.double().cpu(), ref.double().cpu()
    return ((a - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()


def last_launch(L):
    """-> dict of the launch policy's choice for the most recent plane-format launch (ldetr_p3_last_launch)."""
    info = (ctypes.c_int32 * 10)()
    _core().check(L.ldetr_p3_last_launch(info), 'last_launch')
    keys = ('kind', 'bm', 'bn', 'nw', 'splitk', 'xm', 'xn', 'ncls', 'tn_splitk', 'grid')
    return dict(zip(keys, list(info)))


SWEEP_FORCED = 'P3_' in os.environ.get('LDETR_DEBUG', '')      # a development sweep is forcing tiles / prefetch / split-K / pairing (LDETR_DEBUG="P3_TILE=..")


def expect_launch(L, what, **want):
    """The template / policy branch a shape is meant to exercise (skipped under the development sweeps' forcing switches)."""
    if SWEEP_FORCED:
        return
    got = last_launch(L)
    for k, v in want.items():
        ok = v(got[k]) if callable(v) else got[k] == v
        assert ok, f'{what}: launch policy chose {got}, expected {k} = {v if not callable(v) else "<predicate>"}'


def run_fwd(L, xp, N, H, W, Ci, wp, Co, KH, KW, s, pad, ep, yp, yf):
    core = _core()
    core.check(L.ldetr_p3_conv2d_fwd(core.ptr(xp), N, H, W, Ci, core.ptr(wp), Co, KH, KW, s, pad, ctypes.byref(ep) if ep is not None else None,
                                     core.ptr(yp), core.ptr(yf), core.stream()), 'p3_conv2d_fwd')


# ------------------------------------------------------------------------------------------ format
def test_p3_split_merge_is_exact_for_every_fp32_class(dev, L):
    torch.manual_seed(0)
    x = torch.randn(257, 64, device=dev) * torch.exp(4 * torch.randn(257, 64, device=dev))
    fmax = torch.finfo(torch.float32).max
    tiny = torch.finfo(torch.float32).tiny
    # (fp32 subnormals below bf16's own subnormal range, 2^-133, have no plane representation: not part of the claim -- the matrix pipe flushes
    # subnormal operands anyway; 2^-129 is representable)
    special = torch.tensor([0.0, -0.0, float('inf'), -float('inf'), float('nan'), fmax, -fmax, fmax * (1 - 2 ** -20), 3.3895e38, -3.39e38,
                            tiny, -tiny, tiny / 8, 1.0, -1.0, 2 ** -126 * 1.5, 65504.0, 1 + 2 ** -23, 1 - 2 ** -24], device=dev)
    x.view(-1)[:special.numel()] = special
    x.view(-1)[1000:1000 + special.numel()] = special.flip(0)
    p = p3_split(L, x)
    y = p3_merge(L, p, *x.shape)
    xb, yb = x.view(torch.int32), y.view(torch.int32)
    nan = torch.isnan(x)
    assert torch.equal(torch.isnan(y), nan), 'NaN positions changed'
    nz = ~nan & (x != 0)                                  # (-0.0 comes back as +0.0: -0 + 0 + 0 in round-to-nearest; the value is the same)
    assert torch.equal(xb[nz], yb[nz]), 'split -> merge is not the identity on the bit patterns'
    assert bool((y[~nan & (x == 0)] == 0).all())
    # a strided source (row pitch > C) converts the same way
    big = torch.zeros(257, 96, device=dev); big[:, :64] = x
    assert torch.equal(p3_split(L, big[:, :64]), p)


# ------------------------------------------------------------------------------------------ forward
FWD_CASES = [
    # N, H, W, Ci, Co, k, stride, pad, full epilogue
    (1, 8, 8, 32, 64, 1, 1, 0, False),
    (2, 8, 8, 64, 64, 3, 1, 1, False),       # patch kernel, 8x8 images (two images per tile)
    (2, 9, 9, 64, 72, 3, 1, 1, True),        # ragged: gather kernel, Cout not a multiple of the tile
    (2, 16, 16, 64, 128, 3, 2, 1, True),     # stride 2
    (3, 16, 16, 128, 256, 1, 2, 0, True),    # strided 1x1 (downsample)
    (4, 32, 32, 128, 128, 3, 1, 1, True),    # patch kernel 8 x 16
    (3, 16, 32, 64, 96, 3, 1, 1, True),      # non-square image on the patch kernel
    (2, 4, 4, 512, 64, 3, 1, 1, True),       # 4 x 4 images: many images per tile
    (16, 8, 8, 512, 512, 3, 1, 1, True),     # split-K (few tiles, long reduction)
    (16, 16, 16, 1024, 256, 1, 1, 0, True),  # split-K 1x1
    (5, 12, 20, 96, 40, 3, 1, 1, True),      # nothing a power of two
]

# The shapes bench.py's B=16 / 256x256 step launches (profiles/r04h_engine_shapes.txt) with the policy branch each one must reach: the 128x128 /
# 8-wave tile on the large forward grids, the XCD array over >= 1024 tiles, the 256- vs 512-slot split-K targets, the patch kernel.
BENCH_FWD_CASES = [
    # N, H, W, Ci, Co, k, stride, pad, expected launch
    (16, 64, 64, 64, 256, 1, 1, 0, dict(kind=1, bm=128, bn=128, nw=8, splitk=1)),       # layer1 expand: M = 65536 -> 128x128 tile, 1024 tiles over the XCD array
    (16, 64, 64, 256, 64, 1, 1, 0, dict(kind=1, bm=64, bn=64, splitk=1, grid=lambda g: g >= 1024)),   # layer1 reduce: N = 64 keeps the 64x64 tile, 1024 tiles
    (16, 64, 64, 64, 64, 3, 1, 1, dict(kind=2)),                                         # layer1 3x3: patch kernel
    (16, 64, 64, 256, 512, 1, 2, 0, dict(kind=1, bm=128, bn=128, nw=8)),                 # layer2 downsample: strided 1x1, M = 16384, N = 512
    (16, 32, 32, 512, 128, 1, 1, 0, dict(kind=1, bm=64, bn=64, splitk=lambda k: k >= 1)), # layer2 reduce
    (16, 32, 32, 128, 128, 3, 1, 1, dict(kind=2)),                                       # layer2 3x3
    (16, 32, 32, 128, 512, 1, 1, 0, dict(kind=1, bm=128, bn=128, nw=8)),                 # layer2 expand: M = 16384, N >= 256, K <= 512
    (16, 16, 16, 1024, 256, 1, 1, 0, dict(kind=1, bm=64, bn=64, splitk=lambda k: k > 1)), # layer3 reduce: 256 tiles -> split-K
    (16, 16, 16, 256, 1024, 1, 1, 0, dict(kind=1, bm=64, bn=64)),                        # layer3 expand
    (16, 8, 8, 2048, 512, 1, 1, 0, dict(kind=1, bm=64, bn=64, splitk=lambda k: k > 1)),  # layer4 reduce: 128 tiles, K = 2048
]


@pytest.mark.parametrize('case', BENCH_FWD_CASES, ids=lambda c: 'N{}_{}x{}_{}to{}_k{}s{}'.format(*c[:7]))
def test_p3_conv_forward_bench_shapes_vs_float64(dev, L, case):
    """Forward convolutions at the exact geometry of the headline bench (B=16, 256x256 backgrounds) against float64 conv2d, with FrozenBN + residual
    + ReLU in the epilogue, and the launch policy's branch asserted -- the shapes below M = 4096 of FWD_CASES never reach these templates."""
    N, H, W, Ci, Co, k, s, pad, want = case
    torch.manual_seed(11)
    x = torch.randn(N, H, W, Ci, device=dev) * torch.exp(0.5 * torch.randn(N, H, W, Ci, device=dev))
    w = torch.randn(Co, k, k, Ci, device=dev) / (k * Ci ** 0.5)
    OH, OW = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
    xp, wp = p3_split(L, x.reshape(-1, Ci)), p3_split(L, w.reshape(Co, -1))
    yf = torch.empty(N, OH, OW, Co, device=dev)
    yp = torch.empty(N * OH * OW * Co * 6, dtype=torch.uint8, device=dev)
    sc = torch.rand(Co, device=dev) + 0.5; sh = torch.randn(Co, device=dev); res = torch.randn(N, OH, OW, Co, device=dev)
    resp = p3_split(L, res.reshape(-1, Co))
    run_fwd(L, xp, N, H, W, Ci, wp, Co, k, k, s, pad, epilogue(sc, sh, residual_p3=resp, relu=True), yp, yf)
    expect_launch(L, 'forward', **want)
    torch.cuda.synchronize()
    tail = lambda v: torch.relu(v * sc.double().cpu() + sh.double().cpu() + res.double().cpu())
    ref, ref32 = tail(conv_ref64(x, w, s, pad)), tail(conv_f32_host(x, w, s, pad).double())
    e, e32 = err(yf, ref), err(ref32, ref)
    assert e <= max(2e-6, 2 * e32), f'forward: {e:.2e} from float64 (host fp32 conv: {e32:.2e})'
    assert torch.equal(p3_merge(L, yp, N * OH * OW, Co), yf.reshape(-1, Co)), 'the P3 output is not the split of the fp32 output'


@pytest.mark.parametrize('case', FWD_CASES, ids=lambda c: 'N{}_{}x{}_{}to{}_k{}s{}'.format(*c[:7]))
def test_p3_conv_forward_vs_float64(dev, L, case):
    N, H, W, Ci, Co, k, s, pad, full = case
    torch.manual_seed(1)
    x = torch.randn(N, H, W, Ci, device=dev) * torch.exp(torch.randn(N, H, W, Ci, device=dev))
    w = torch.randn(Co, k, k, Ci, device=dev) / (k * Ci ** 0.5)
    OH, OW = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
    xp, wp = p3_split(L, x.reshape(-1, Ci)), p3_split(L, w.reshape(Co, -1))
    yf = torch.empty(N, OH, OW, Co, device=dev)
    yp = torch.empty(N * OH * OW * Co * 6, dtype=torch.uint8, device=dev)
    ref = conv_ref64(x, w, s, pad)
    ref32 = conv_f32_host(x, w, s, pad).double()
    ep = None
    if full:
        sc = torch.rand(Co, device=dev) + 0.5; sh = torch.randn(Co, device=dev); res = torch.randn(N, OH, OW, Co, device=dev)
        resp = p3_split(L, res.reshape(-1, Co))
        ep = epilogue(sc, sh, residual_p3=resp, relu=True)
        tail = lambda v: torch.relu(v * sc.double().cpu() + sh.double().cpu() + res.double().cpu())
        ref, ref32 = tail(ref), tail(ref32)
    run_fwd(L, xp, N, H, W, Ci, wp, Co, k, k, s, pad, ep, yp, yf)
    torch.cuda.synchronize()
    e, e32 = err(yf, ref), err(ref32, ref)
    assert e <= max(2e-6, 2 * e32), f'forward: {e:.2e} from float64 (host fp32 conv: {e32:.2e})'
    assert torch.equal(p3_merge(L, yp, N * OH * OW, Co), yf.reshape(-1, Co)), 'the P3 output is not the split of the fp32 output'
    # fp32 residual instead of the plane one: same values
    if full:
        yf2 = torch.empty_like(yf)
        ep2 = epilogue(sc, sh, residual_f32=res, relu=True)
        run_fwd(L, xp, N, H, W, Ci, wp, Co, k, k, s, pad, ep2, None, yf2)
        assert torch.equal(yf2, yf)


# ------------------------------------------------------------------------------------------ backward
BWD_CASES = [
    (1, 8, 8, 32, 32, 1, 1, 0),
    (2, 8, 8, 64, 64, 3, 1, 1),
    (2, 9, 9, 64, 96, 3, 1, 1),
    (2, 16, 16, 64, 128, 3, 2, 1),
    (3, 16, 16, 128, 256, 1, 2, 0),
    (4, 32, 32, 128, 128, 3, 1, 1),
    (3, 16, 32, 64, 96, 3, 1, 1),
    (16, 8, 8, 512, 512, 3, 1, 1),
    (16, 16, 16, 1024, 256, 1, 1, 0),
    (2, 14, 10, 64, 64, 3, 2, 1),            # stride 2 on a ragged grid: parity classes of different sizes
    # bench.py's B=16 / 256x256 geometry (profiles/r04h_engine_shapes.txt): the paired launches that dominate the step
    (16, 64, 64, 64, 256, 1, 1, 0),          # M = 65536: weight-gradient pixel slices over the XCDs
    (16, 64, 64, 256, 64, 1, 1, 0),
    (16, 64, 64, 256, 512, 1, 2, 0),         # strided 1x1: four parity classes, three of them without taps
    (16, 64, 64, 128, 128, 3, 2, 1),         # layer2's strided 3x3
    (16, 32, 32, 512, 128, 1, 1, 0),
    (16, 32, 32, 128, 128, 3, 1, 1),         # paired patch kernel
    (16, 16, 16, 1024, 256, 1, 1, 0),
    (16, 8, 8, 2048, 512, 1, 1, 0),
]
BENCH_PAIR_KIND = {(16, 64, 64, 64, 256, 1, 1, 0): 4, (16, 64, 64, 256, 64, 1, 1, 0): 4, (16, 64, 64, 256, 512, 1, 2, 0): 4, (16, 64, 64, 128, 128, 3, 2, 1): 4,
                   (16, 32, 32, 512, 128, 1, 1, 0): 4, (16, 32, 32, 128, 128, 3, 1, 1): 5, (16, 16, 16, 1024, 256, 1, 1, 0): 4, (16, 8, 8, 2048, 512, 1, 1, 0): 4}


@pytest.mark.parametrize('case', BWD_CASES, ids=lambda c: 'N{}_{}x{}_{}to{}_k{}s{}'.format(*c[:7]))
def test_p3_conv_backward_vs_float64(dev, L, case):
    core = _core()
    N, H, W, Ci, Co, k, s, pad = case
    torch.manual_seed(2)
    OH, OW = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
    x = torch.randn(N, H, W, Ci, device=dev)
    w = torch.randn(Co, k, k, Ci, device=dev) / (k * Ci ** 0.5)
    dy = torch.randn(N, OH, OW, Co, device=dev) * torch.exp(torch.randn(N, OH, OW, Co, device=dev))
    sc = torch.rand(Co, device=dev) + 0.5                                   # FrozenBN's factor of the output gradient
    res = torch.randn(N, H, W, Ci, device=dev)

    def grads(dt):
        xd = x.cpu().to(dt).permute(0, 3, 1, 2).requires_grad_(True); wd = w.cpu().to(dt).permute(0, 3, 1, 2).requires_grad_(True)
        y = F.conv2d(xd, wd, stride=s, padding=pad)
        gx, gw = torch.autograd.grad(y, (xd, wd), (dy.cpu().to(dt) * sc.cpu().to(dt)).permute(0, 3, 1, 2))
        return gx.permute(0, 2, 3, 1).double(), gw.permute(0, 2, 3, 1).double()

    gx, gw = grads(torch.float64)
    gx32, gw32 = grads(torch.float32)
    dyp, xp, resp = p3_split(L, dy.reshape(-1, Co)), p3_split(L, x.reshape(-1, Ci)), p3_split(L, res.reshape(-1, Ci))
    wb = p3_weight_bwd(L, w, sc)
    # data gradient with the ReLU mask of the conv's input and a pass-through gradient added (the bottleneck's fan-in)
    ep = epilogue(residual_p3=resp, mask_p3=xp)
    dxf = torch.empty(N, H, W, Ci, device=dev); dxp = torch.empty(N * H * W * Ci * 6, dtype=torch.uint8, device=dev)
    core.check(L.ldetr_p3_conv2d_bwd_data(core.ptr(dyp), N, OH, OW, Co, core.ptr(wb), Ci, k, k, s, pad, H, W, ctypes.byref(ep), core.ptr(dxp), core.ptr(dxf), core.stream()), 'bwd_data')
    keep = (x > 0).cpu()
    ref = torch.where(keep, gx + res.double().cpu(), torch.zeros_like(gx))
    ref32 = torch.where(keep, gx32 + res.double().cpu(), torch.zeros_like(gx))
    e, e32 = err(dxf, ref), err(ref32, ref)
    assert e <= max(3e-6, 2 * e32), f'data gradient: {e:.2e} from float64 (host fp32: {e32:.2e})'
    assert torch.equal(p3_merge(L, dxp, N * H * W, Ci), dxf.reshape(-1, Ci))
    # weight gradient, accumulated onto an existing buffer (the flat .grad)
    dw0 = torch.randn(Co, k, k, Ci, device=dev); dw = dw0.clone()
    core.check(L.ldetr_p3_conv2d_bwd_weight(core.ptr(xp), N, H, W, Ci, core.ptr(dyp), Co, k, k, s, pad, core.ptr(sc), core.ptr(dw), core.stream()), 'bwd_weight')
    torch.cuda.synchronize()
    e, e32 = err((dw - dw0), gw), err(gw32, gw)
    assert e <= max(3e-6, 3 * e32), f'weight gradient: {e:.2e} from float64 (host fp32: {e32:.2e})'   # (dw - dw0 itself rounds at |dw0|'s ulp)
    # both gradients as ONE launch (ldetr_p3_conv2d_bwd_pair): the same blocks in one grid -> the same dx bit for bit, dw up to the order of its atomics
    dxp2 = torch.zeros_like(dxp); dw2 = dw0.clone(); nl = ctypes.c_int(-1)
    core.check(L.ldetr_p3_conv2d_bwd_pair(core.ptr(dyp), N, OH, OW, Co, core.ptr(wb), core.ptr(xp), Ci, k, k, s, pad, H, W, ctypes.byref(ep), core.ptr(dxp2), None,
                                          core.ptr(sc), core.ptr(dw2), ctypes.byref(nl), core.stream()), 'bwd_pair')
    torch.cuda.synchronize()
    dbg = os.environ.get('LDETR_DEBUG', '')
    forced = any(k in dbg for k in ('P3_TILE', 'P3_WTILE', 'P3_PF', 'P3_WPF', 'P3_NST', 'P3_PAIR=0'))
    want = 2 if forced else 1      # the development switches fall back to the two launches
    assert nl.value == want, f'expected {want} launch(es) for the pair, got {nl.value}'
    if tuple(case) in BENCH_PAIR_KIND:
        expect_launch(L, 'paired backward', kind=BENCH_PAIR_KIND[tuple(case)], ncls=(s * s if (s > 1 and BENCH_PAIR_KIND[tuple(case)] == 4) else 1))
    assert torch.equal(dxp2, dxp), 'paired launch: data gradient differs from the separate launch'
    e = err((dw2 - dw0), gw)
    assert e <= max(3e-6, 3 * e32), f'paired launch: weight gradient {e:.2e} from float64 (host fp32: {e32:.2e})'


def test_p3_weight_images_in_one_launch_equal_the_single_tensor_entry_points(dev, L):
    """ldetr_p3_weight_prep (all conv weights of a module per optimiser step) == ldetr_p3_split_f32 / ldetr_p3_weight_bwd per tensor."""
    core = _core()
    torch.manual_seed(3)
    shapes = [(64, 1, 1, 64), (64, 3, 3, 64), (256, 1, 1, 64), (128, 3, 3, 128), (40, 3, 3, 24)]
    ws = [torch.randn(s, device=dev) for s in shapes]
    scs = [torch.rand(s[0], device=dev) + 0.5 if i % 2 == 0 else None for i, s in enumerate(shapes)]
    total = sum(w.numel() * 6 for w in ws)
    fwd = torch.zeros(total, dtype=torch.uint8, device=dev); bwd = torch.zeros(total, dtype=torch.uint8, device=dev)
    rows, off, blk = [], 0, 0
    for w, sc in zip(ws, scs):
        O, KH, KW, I = w.shape
        rows.append([w.data_ptr(), sc.data_ptr() if sc is not None else 0, fwd.data_ptr() + off, bwd.data_ptr() + off, O, KH * KW, I, blk])
        off += w.numel() * 6; blk += (w.numel() // 8 + 255) // 256
    table = torch.tensor(rows, dtype=torch.int64, device=dev)
    core.check(L.ldetr_p3_weight_prep(core.ptr(table), len(ws), blk, core.stream()), 'prep')
    off = 0
    for w, sc in zip(ws, scs):
        n = w.numel() * 6
        assert torch.equal(fwd[off:off + n], p3_split(L, w.reshape(w.shape[0], -1)))
        assert torch.equal(bwd[off:off + n], p3_weight_bwd(L, w, sc))
        off += n


# ------------------------------------------------------------------------------------------ non-finite operands
def _classes(t):
    t = t.cpu()
    return torch.isnan(t), torch.isposinf(t), torch.isneginf(t)


@pytest.mark.parametrize('geom', [(4, 16, 16, 64, 64, 3, 1, 1), (4, 16, 16, 128, 64, 1, 1, 0), (2, 16, 16, 64, 64, 3, 2, 1), (8, 8, 8, 512, 128, 3, 1, 1)],
                         ids=['patch3x3', 'gather1x1', 'stride2', 'splitk'])
def test_p3_non_finite_operands_give_the_fp32_convolutions_classes(dev, L, geom):
    """+-Inf, NaN and FLT_MAX planted in activations and weights: every output must be NaN / +Inf / -Inf / finite exactly where torch's fp32
    convolution of the same operands is (the three-way split is exact for finite values only: Inf = Inf + NaN + NaN as planes would turn
    every Inf into NaN, and training_loop.py:308 maps NaN -> 0 but +-Inf -> +-1e5), clean outputs keep their fp32-equivalent values, and the
    sanitised gradients a step would apply are the reference's."""
    core = _core()
    N, H, W, Ci, Co, k, s, pad = geom
    torch.manual_seed(4)
    x = torch.randn(N, H, W, Ci, device=dev); w = torch.randn(Co, k, k, Ci, device=dev) / (k * Ci ** 0.5)
    fmax = torch.finfo(torch.float32).max
    x[0, 3, 4, 5] = float('inf'); x[1, 7, 7, 9] = -float('inf'); x[2 % N, 0, 0, 0] = float('nan'); x[3 % N, 5, 2, 17] = fmax; x[1, 2, 3, 4] = -fmax
    c = k // 2        # weights: the centre tap only -- it never meets the zero padding, where 0 x Inf depends on the convolution algorithm
    w[3, c, c, 7] = float('inf'); w[5, c, c, 1] = float('nan'); w[9, c, c, 2] = fmax
    x[..., 2] *= 0.1  # FLT_MAX x |x| < 1 stays finite: a product that overflows to -Inf on its own meets a +Inf partial sum as NaN or as +Inf depending on
    #                   where an fp32 evaluation splits the reduction (one fma chain absorbs it, split-K slices do not) -- not a class this test can pin
    OH, OW = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
    xp, wp = p3_split(L, x.reshape(-1, Ci)), p3_split(L, w.reshape(Co, -1))
    assert torch.equal(p3_merge(L, xp, N * H * W, Ci)[~torch.isnan(x.reshape(-1, Ci))], x.reshape(-1, Ci)[~torch.isnan(x.reshape(-1, Ci))])
    yf = torch.empty(N, OH, OW, Co, device=dev); yp = torch.empty(N * OH * OW * Co * 6, dtype=torch.uint8, device=dev)
    run_fwd(L, xp, N, H, W, Ci, wp, Co, k, k, s, pad, None, yp, yf)
    torch.cuda.synchronize()
    ref = conv_f32_host(x, w, s, pad)
    got = yf.cpu()
    for name, a, b in zip(('NaN', '+Inf', '-Inf'), _classes(got), _classes(ref)):
        assert torch.equal(a, b), f'forward: {name} positions differ from the fp32 convolution ({int(a.sum())} vs {int(b.sum())})'
    fin = torch.isfinite(ref)
    assert fin.any() and (~fin).any()
    big = fin & (ref.abs() > 1e30)      # sums next to the overflow threshold depend on the summation order in ANY fp32 evaluation
    ok = fin & ~big
    assert (got[ok] - ref[ok]).abs().max() <= 2e-5 * ref[ok].abs().max()
    pm = p3_merge(L, yp, N * OH * OW, Co).cpu().reshape(got.shape)
    assert torch.equal(torch.isnan(pm), torch.isnan(got)) and torch.equal(pm[~torch.isnan(pm)], got[~torch.isnan(got)]), 'P3 output differs from the fp32 output'

    # backward: a poisoned output gradient
    dy = torch.randn(N, OH, OW, Co, device=dev)
    # (interior pixels: at the border the zero padding meets the poison, and 0 x NaN there is implementation-defined -- a direct convolution
    # skips the padding, an implicit GEMM, like this engine, multiplies it)
    dy[0, 1, 1, 3] = float('inf'); dy[1, 2, 2, 5] = float('nan'); dy[2 % N, 3, 3, 8] = -float('inf')
    xc = torch.randn(N, H, W, Ci, device=dev); wc = torch.randn(Co, k, k, Ci, device=dev) / (k * Ci ** 0.5)     # clean operands
    xd = xc.cpu().permute(0, 3, 1, 2).requires_grad_(True); wd = wc.cpu().permute(0, 3, 1, 2).requires_grad_(True)
    gx, gw = torch.autograd.grad(F.conv2d(xd, wd, stride=s, padding=pad), (xd, wd), dy.cpu().permute(0, 3, 1, 2))
    gx, gw = gx.permute(0, 2, 3, 1), gw.permute(0, 2, 3, 1)
    dyp, xcp = p3_split(L, dy.reshape(-1, Co)), p3_split(L, xc.reshape(-1, Ci))
    wb = p3_weight_bwd(L, wc)
    dxf = torch.empty(N, H, W, Ci, device=dev)
    core.check(L.ldetr_p3_conv2d_bwd_data(core.ptr(dyp), N, OH, OW, Co, core.ptr(wb), Ci, k, k, s, pad, H, W, None, None, core.ptr(dxf), core.stream()), 'bwd_data')
    dw = torch.zeros(Co, k, k, Ci, device=dev)
    core.check(L.ldetr_p3_conv2d_bwd_weight(core.ptr(xcp), N, H, W, Ci, core.ptr(dyp), Co, k, k, s, pad, None, core.ptr(dw), core.stream()), 'bwd_weight')
    torch.cuda.synchronize()
    for what, a, b in (('data gradient', dxf.cpu(), gx), ('weight gradient', dw.cpu(), gw)):
        for name, ca, cb in zip(('NaN', '+Inf', '-Inf'), _classes(a), _classes(b)):
            assert torch.equal(ca, cb), f'{what}: {name} positions differ from fp32 autograd ({int(ca.sum())} vs {int(cb.sum())})'
        f = torch.isfinite(b)
        assert (a[f] - b[f]).abs().max() <= 3e-5 * b[f].abs().max(), what
        # what the optimiser would see (training_loop.py:308)
        assert torch.allclose(torch.nan_to_num(a, nan=0, posinf=1e5, neginf=-1e5), torch.nan_to_num(b, nan=0, posinf=1e5, neginf=-1e5), rtol=3e-5, atol=3e-5 * b[f].abs().max().item())


def test_p3_relu_and_mask_follow_aten_on_nan(dev, L):
    """relu(NaN) = NaN and its gradient mask (y > 0) is false for NaN, as ATen's threshold / threshold_backward."""
    core = _core()
    N, H, W, Ci, Co = 1, 8, 8, 32, 32
    torch.manual_seed(5)
    x = torch.randn(N, H, W, Ci, device=dev); w = torch.randn(Co, 1, 1, Ci, device=dev)
    x[0, 2, 2, 3] = float('nan')
    xp, wp = p3_split(L, x.reshape(-1, Ci)), p3_split(L, w.reshape(Co, -1))
    yf = torch.empty(N, H, W, Co, device=dev)
    run_fwd(L, xp, N, H, W, Ci, wp, Co, 1, 1, 1, 0, epilogue(relu=True), None, yf)
    ref = torch.relu(conv_f32_host(x, w, 1, 0))
    assert torch.equal(torch.isnan(yf.cpu()), torch.isnan(ref)) and torch.isnan(ref).sum() == Co
    # mask from a P3 tensor with NaN / Inf / -0 / tiny entries
    m = torch.randn(N, H, W, Ci, device=dev)
    m.view(-1)[:6] = torch.tensor([float('nan'), float('inf'), -float('inf'), -0.0, 0.0, 1e-30], device=dev)
    mp = p3_split(L, m.reshape(-1, Ci))
    dy = torch.randn(N, H, W, Co, device=dev); dyp = p3_split(L, dy.reshape(-1, Co))
    wb = p3_weight_bwd(L, w)
    dx = torch.empty(N, H, W, Ci, device=dev)
    core.check(L.ldetr_p3_conv2d_bwd_data(core.ptr(dyp), N, H, W, Co, core.ptr(wb), Ci, 1, 1, 1, 0, H, W, ctypes.byref(epilogue(mask_p3=mp)), None, core.ptr(dx), core.stream()), 'bwd')
    torch.cuda.synchronize()
    keep = (m > 0).cpu()
    assert torch.equal(dx.cpu() != 0, keep & (dx.cpu() != 0)) and (dx.cpu()[keep] != 0).all(), 'mask is not (m > 0)'


# ------------------------------------------------------------------------------------------ the trunk on both engines
def test_resnet_trunk_plane_path_equals_fp32_engine_path(dev):
    """ResNet50Body forward + backward on P3 activations (default) against the same module on the fp32 engine (LDETR_TRUNK_P3=0): both are
    fp32-equivalent evaluations of the reference's trunk, so outputs agree to fp32 rounding and gradients up to flipped ReLU units."""
    from layoutdetr_amd.training.detr_backbone import ResNet50Body
    torch.manual_seed(6)
    body = ResNet50Body().to(dev).train()
    for p in body.parameters():
        p.requires_grad_(True)
    img = torch.randn(2, 3, 64, 64, device=dev)
    outs = {}
    prev = os.environ.get('LDETR_TRUNK_P3')
    try:
        for mode in ('1', '0'):
            os.environ['LDETR_TRUNK_P3'] = mode
            body.zero_grad(set_to_none=True)
            y = body(img)
            g = torch.randn(y.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(7))
            y.backward(g)
            outs[mode] = (y.detach().clone(), {n: p.grad.detach().clone() for n, p in body.named_parameters() if p.grad is not None})
    finally:
        if prev is None:
            os.environ.pop('LDETR_TRUNK_P3', None)
        else:
            os.environ['LDETR_TRUNK_P3'] = prev
    y1, g1 = outs['1']; y0, g0 = outs['0']
    assert err(y1, y0) <= 5e-6, f'trunk output: {err(y1, y0):.2e}'
    assert set(g1) == set(g0) and len(g1) > 50
    errs = sorted(err(g1[n], g0[n]) for n in g1)
    med, p90 = errs[len(errs) // 2], errs[int(len(errs) * 0.9)]
    assert med <= 1e-4 and p90 <= 5e-3, f'trunk gradients: median {med:.2e}, p90 {p90:.2e}, worst {errs[-1]:.2e}'


def test_dual_trunk_forward_equals_two_forwards(dev):
    """detr_backbone.dual_trunk_forward (every plane-format conv of G's and D's trunk as one grouped launch, each module's autograd graph replayed around the
    precomputed activations) against the two ordinary forwards: the same kernels on the same tiles -> identical outputs, and the same gradients up to the
    order of the weight gradients' atomics."""
    from layoutdetr_amd.training.detr_backbone import ResNet50Body, dual_trunk_forward
    torch.manual_seed(8)
    ba, bb = ResNet50Body().to(dev).train(), ResNet50Body().to(dev).train()
    for p in list(ba.parameters()) + list(bb.parameters()):
        p.requires_grad_(True)
    img = torch.randn(2, 3, 64, 64, device=dev)
    ga = torch.randn(2, 2, 2, 2048, device=dev); gb = torch.randn(2, 2, 2, 2048, device=dev)
    ya, yb = ba(img), bb(img)
    (ya * ga).sum().backward(); (yb * gb).sum().backward()
    ref = {id(p): p.grad.clone() for p in list(ba.parameters()) + list(bb.parameters()) if p.grad is not None}
    for p in list(ba.parameters()) + list(bb.parameters()):
        p.grad = None
    da, db = dual_trunk_forward(ba, bb, img, img)
    assert torch.equal(da, ya) and torch.equal(db, yb), 'grouped forward differs from the two separate forwards'
    (da * ga).sum().backward()          # the two backward passes are independent (they run in different phases of the step)
    (db * gb).sum().backward()
    n = 0
    for p in list(ba.parameters()) + list(bb.parameters()):
        if id(p) in ref:
            e = err(p.grad, ref[id(p)]); n += 1
            assert e <= 2e-5, f'gradient of a {tuple(p.shape)} parameter: {e:.2e}'
    assert n > 100
