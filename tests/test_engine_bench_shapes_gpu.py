"""The fp32-operand engine (csrc/gemm_conv.hip, conv_c32.hip, wgrad_smallc.hip) at the EXACT shapes and tile policies bench.py's headline step
launches (B = 16 per GPU, 256 x 256 backgrounds: profiles/r06_engine_shapes.txt), each against float64 at <= 2e-5 with the launched kernel
template asserted (ldetr_engine_last_launch) -- so that a policy change cannot silently move a bench shape onto a template no parity case
reaches, and the flip-tolerant full-iteration gate at B = 16 is not the only guard of these kernels (tests/test_p3_gpu.py does the same for
the plane-format trunk).  Covered: every StyleGAN2 synthesis layer of the 256 x 256 decoder (3x3 and transposed 3x3 + FIR: forward, data
gradient, weight gradient, style / demodulation / bias gradients), the 64-token encoders' projections and feed-forward GEMMs at 1024 and
2048 rows, the paired dX + dW launches of the heads."""
import math
import os

import pytest
import torch

from oracle import ops_ref

pytestmark = pytest.mark.gpu

SWEEP_FORCED = any(k in os.environ.get('LDETR_DEBUG', '') for k in ('FORCE_TILE', 'FORCE_SK', 'SPLIT_BF16', 'GEMM_PAIR', 'SMALL_FAST', 'FAST_LOADS'))


def rel(a, ref):
    a, ref = a.detach().double().cpu(), ref.detach().double().cpu()
    return ((a - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()


class Launches(object):
    """`with Launches() as L:` records (C-ABI entry, launched kernel template) of every engine call inside (hip.core.engine_call)."""

    def __enter__(self):
        from layoutdetr_amd.hip import core
        self.core = core
        self.was = core.PROF.enabled
        core.PROF.enabled = True
        self.start = len(core.PROF.records)
        return self

    def __exit__(self, *exc):
        self.core.PROF.enabled = self.was
        self.seen = [(r[0].replace('ldetr_', '').replace('_f32', ''), r[6]) for r in self.core.PROF.records[self.start:]]
        del self.core.PROF.records[self.start:]
        return False


def expect(seen, want, what):
    """want: {entry: kernel template}; every listed entry must have launched exactly that template (skipped under the development sweeps'
    forcing switches)."""
    got = {}
    for entry, lab in seen:
        got.setdefault(entry, []).append(lab)
    print(f'  {what}: ' + '; '.join(f'{e} -> {", ".join(l)}' for e, l in got.items()))
    if SWEEP_FORCED or want is None:
        return
    for entry, labs in want.items():
        assert entry in got, f'{what}: no {entry} launch recorded ({got})'
        assert got[entry] == list(labs), f'{what}: {entry} launched {got[entry]}, the bench shape is pinned to {list(labs)}'


# the 256 x 256 Decoder of networks_stylegan2.py:482-497 at 16 samples per GPU: (resolution of the layer's INPUT, Cin, Cout, up)
SG2_LAYERS = [
    (4, 512, 512, 1), (4, 512, 512, 2), (8, 512, 512, 1), (8, 512, 512, 2), (16, 512, 512, 1), (16, 512, 256, 2), (32, 256, 256, 1),
    (32, 256, 128, 2), (64, 128, 128, 1), (64, 128, 64, 2), (128, 64, 64, 1), (128, 64, 32, 2), (256, 32, 32, 1),
]
# what the launch policy chooses for each layer at B = 16 (recorded on an MI355X; a policy change must update this table AND keep the parity below green)
SG2_EXPECT = {
    (4, 512, 512, 1): {'conv2d_fwd': ['gemm_f32_kernel<64,64,32,4w,FAST,SPLIT> splitK=12'], 'conv2d_bwd_data': ['gemm_f32_kernel<64,64,32,4w,FAST,SPLIT> splitK=12'], 'conv2d_bwd_weight': ['gemm_f32_kernel<64,64,32,4w>']},
    (4, 512, 512, 2): {'conv_transpose2d_fwd': ['gemm_f32_kernel<64,64,32,4w,FAST,SPLIT>'], 'conv_transpose2d_bwd_data': ['gemm_f32_kernel<64,64,32,4w,FAST,SPLIT> splitK=12'], 'conv_transpose2d_bwd_weight': ['gemm_f32_kernel<64,64,32,4w>']},
    (8, 512, 512, 1): {'conv2d_fwd': ['gemm_f32_kernel<128,64,32,4w,FAST,SPLIT> splitK=8'], 'conv2d_bwd_data': ['gemm_f32_kernel<128,64,32,4w,FAST,SPLIT> splitK=8'], 'conv2d_bwd_weight': ['gemm_f32_kernel<128,64,32,4w,FAST,SPLIT> splitK=2']},
    (8, 512, 512, 2): {'conv_transpose2d_fwd': ['gemm_f32_kernel<64,64,32,4w,FAST,SPLIT>'], 'conv_transpose2d_bwd_data': ['gemm_f32_kernel<128,64,32,4w,FAST,SPLIT> splitK=8'], 'conv_transpose2d_bwd_weight': ['gemm_f32_kernel<128,64,32,4w,FAST,SPLIT> splitK=2']},
    (16, 512, 512, 1): {'conv2d_fwd': ['gemm_f32_kernel<128,64,32,4w,FAST,SPLIT> splitK=2'], 'conv2d_bwd_data': ['gemm_f32_kernel<128,64,32,4w,FAST,SPLIT> splitK=2'], 'conv2d_bwd_weight': ['gemm_f32_kernel<128,128,32,4w,FAST,SPLIT> splitK=4']},
    (16, 512, 256, 2): {'conv_transpose2d_fwd': ['gemm_f32_kernel<128,64,32,4w,FAST,SPLIT>'], 'conv_transpose2d_bwd_data': ['gemm_f32_kernel<128,64,32,4w,FAST,SPLIT> splitK=2'], 'conv_transpose2d_bwd_weight': ['gemm_f32_kernel<128,128,32,4w,FAST,SPLIT> splitK=8']},
    (32, 256, 256, 1): {'conv2d_fwd': ['gemm_f32_kernel<128,64,32,4w,FAST,SPLIT>'], 'conv2d_bwd_data': ['gemm_f32_kernel<128,64,32,4w,FAST,SPLIT>'], 'conv2d_bwd_weight': ['gemm_f32_kernel<128,128,32,4w,FAST,SPLIT> splitK=16']},
    (32, 256, 128, 2): {'conv_transpose2d_fwd': ['gemm_f32_kernel<128,128,32,4w,FAST,SPLIT>'], 'conv_transpose2d_bwd_data': ['gemm_f32_kernel<128,64,32,4w,FAST,SPLIT>'], 'conv_transpose2d_bwd_weight': ['gemm_f32_kernel<128,128,32,4w,FAST,SPLIT> splitK=32']},
    (64, 128, 128, 1): {'conv2d_fwd': ['gemm_f32_kernel<128,128,32,4w,FAST,SPLIT>'], 'conv2d_bwd_data': ['gemm_f32_kernel<128,128,32,4w,FAST,SPLIT>'], 'conv2d_bwd_weight': ['gemm_f32_kernel<128,128,32,4w,FAST,SPLIT> splitK=64']},
    (64, 128, 64, 2): {'conv_transpose2d_fwd': ['gemm_f32_kernel<128,64,32,4w,FAST,SPLIT>'], 'conv_transpose2d_bwd_data': ['gemm_f32_kernel<128,128,32,4w,FAST,SPLIT>'], 'conv_transpose2d_bwd_weight': ['gemm_f32_kernel<64,64,32,4w,FAST,SPLIT> splitK=48']},
    (128, 64, 64, 1): {'conv2d_fwd': ['gemm_f32_kernel<128,64,32,4w,FAST,SPLIT>'], 'conv2d_bwd_data': ['gemm_f32_kernel<128,64,32,4w,FAST,SPLIT>'], 'conv2d_bwd_weight': ['gemm_f32_kernel<64,64,32,4w,FAST,SPLIT> splitK=80']},
    (128, 64, 32, 2): {'conv_transpose2d_fwd': ['gemm_f32_kernel<256,32,32,4w,FAST,SPLIT>'], 'conv_transpose2d_bwd_data': ['gemm_f32_kernel<128,64,32,8w>'], 'conv_transpose2d_bwd_weight': ['gemm_f32_kernel<64,64,32,4w,FAST,SPLIT> splitK=80']},
    (256, 32, 32, 1): {'conv2d_fwd': ['conv3x3_c32_split_kernel'], 'conv2d_bwd_data': ['conv3x3_c32_split_kernel'], 'conv2d_bwd_weight': ['wgrad_c32_3x3_kernel']},
}


@pytest.mark.parametrize('R,Ci,Co,up', SG2_LAYERS, ids=lambda v: str(v))
def test_stylegan2_layers_at_bench_shapes_vs_float64(dev, R, Ci, Co, up):
    """One synthesis layer (modulate -> 3x3 conv | transposed conv + 4x4 FIR -> demodulate -> bias -> lrelu * sqrt 2; networks_stylegan2.py:30-75,
    307-326) at B = 16 and the decoder's own geometry against the oracle's non-fused formulation evaluated in float64."""
    from layoutdetr_amd.hip import modconv
    B = 16
    torch.manual_seed(300 + R + up)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    x = torch.randn(B, Ci, R, R); w = torch.randn(Co, Ci, 3, 3) / math.sqrt(Ci * 9); s = torch.randn(B, Ci) * 0.5 + 1.0
    # a bias of +8 sigma keeps the lrelu of these multi-million-element outputs on one branch (a pre-activation within rounding distance of 0
    # takes the other slope in ANY two fp32 evaluations: test_modulated_conv_layers_vs_oracle covers the kink at small sizes)
    b = torch.randn(Co) * 0.1 + 8.0
    f = ops_ref.setup_filter([1, 3, 3, 1])
    xr, wr, sr, br = [t.double().clone().requires_grad_(True) for t in (x, w, s, b)]
    y = ops_ref.modulated_conv2d(xr, wr, sr, up=up, padding=1, resample_filter=f.double(), demodulate=True, flip_weight=(up == 1))
    y = ops_ref.bias_act(y, br, act='lrelu', gain=math.sqrt(2))
    g = torch.randn(y.shape)
    y.backward(g.double())
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev).requires_grad_(True)
    wd = w.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    sd = s.to(dev).requires_grad_(True); bd = b.to(dev).requires_grad_(True)
    with Launches() as L:
        out = modconv.modconv3x3(xd, wd, sd, bd) if up == 1 else modconv.modconv3x3_up2(xd, wd, sd, bd, f.to(dev))
        out.backward(g.permute(0, 2, 3, 1).contiguous().to(dev))
    torch.cuda.synchronize()
    expect(L.seen, SG2_EXPECT[(R, Ci, Co, up)], f'layer {R}x{R} {Ci}->{Co} up={up}')
    errs = dict(y=rel(out.permute(0, 3, 1, 2), y), dx=rel(xd.grad.permute(0, 3, 1, 2), xr.grad), dw=rel(wd.grad, wr.grad), dstyles=rel(sd.grad, sr.grad),
                dbias=rel(bd.grad, br.grad))
    print('   ', {k: f'{v:.1e}' for k, v in errs.items()})
    for k, v in errs.items():
        assert v <= 2e-5, f'layer {R}x{R} {Ci}->{Co} up={up}: {k} is {v:.2e} from float64'


# Linear layers of the step at B = 16: (rows, in, out, what)
LINEAR_CASES = [
    (1024, 256, 2048, 'encoder linear1 (G, 16 x 64 tokens)'), (1024, 2048, 256, 'encoder linear2'),
    (2048, 256, 2048, "encoder linear1 (D's two passes as one batch: 32 x 64 tokens)"), (2048, 2048, 256, 'encoder linear2, 2048 rows'),
    (1024, 256, 512, 'packed q | k projection'), (1024, 256, 256, 'v / output projection'), (2048, 256, 512, 'packed q | k projection, 2048 rows'),
    (1024, 256, 1536, "grouped K projections of a decoder's six layers"), (2048, 256, 1536, 'grouped K projections, 2048 rows'),
    (144, 3072, 768, 'fc_in layer 0 (16 x 9 elements)'), (144, 768, 768, 'fc_in layer 1'), (144, 768, 256, 'fc_in layer 2'), (288, 3072, 768, 'enc_fc_in layer 0, 32 layouts'),
]
# launches of (forward, dX, dW) in order, parameters WITHOUT a flat gradient buffer (no pairing); the paired dX + dW launch is pinned by FFN_EXPECT and
# by the flat-gradient leg of the test
LINEAR_EXPECT = {
    (1024, 256, 2048): ['gemm_f32_kernel<64,64,32,4w,FAST,SPLIT>', 'gemm_small_kernel<8w,FAST>', 'gemm_small_kernel<4w,FAST>'],
    (1024, 2048, 256): ['gemm_small_kernel<8w,FAST>', 'gemm_f32_kernel<64,64,32,4w,FAST,SPLIT>', 'gemm_small_kernel<4w,FAST>'],
    (2048, 256, 2048): ['gemm_f32_kernel<128,64,32,4w,FAST,SPLIT>', 'gemm_small_kernel<4w,FAST>', 'gemm_small_kernel<4w,FAST>'],
    (2048, 2048, 256): ['gemm_small_kernel<4w,FAST>', 'gemm_f32_kernel<128,64,32,4w,FAST,SPLIT>', 'gemm_small_kernel<4w,FAST>'],
    (1024, 256, 512): ['gemm_small_kernel<4w,FAST>', 'gemm_small_kernel<8w,FAST>', 'gemm_small_kernel<4w,FAST>'],
    (1024, 256, 256): ['gemm_small_kernel<4w,FAST>', 'gemm_small_kernel<4w,FAST>', 'gemm_small_kernel<4w,FAST>'],
    (2048, 256, 512): ['gemm_f32_kernel<64,64,32,4w,FAST,SPLIT>', 'gemm_small_kernel<4w,FAST>', 'gemm_small_kernel<4w,FAST>'],
    (1024, 256, 1536): ['gemm_f32_kernel<64,64,32,4w,FAST,SPLIT>', 'gemm_small_kernel<8w,FAST>', 'gemm_small_kernel<4w,FAST>'],
    (2048, 256, 1536): ['gemm_f32_kernel<64,64,32,4w,FAST,SPLIT>', 'gemm_small_kernel<4w,FAST>', 'gemm_small_kernel<4w,FAST>'],
    (144, 3072, 768): ['gemm_small_kernel<4w,FAST>', 'gemm_small_kernel<4w,FAST>', 'gemm_f32_kernel<64,64,32,4w,FAST,SPLIT>'],
    (144, 768, 768): ['gemm_small_kernel<8w,FAST>', 'gemm_small_kernel<8w,FAST>', 'gemm_small_kernel<4w>'],
    (144, 768, 256): ['gemm_small_kernel<8w,FAST>', 'gemm_small_kernel<4w,FAST>', 'gemm_small_kernel<4w>'],
    (288, 3072, 768): ['gemm_small_kernel<8w,FAST>', 'gemm_small_kernel<4w,FAST>', 'gemm_f32_kernel<64,64,32,4w,FAST,SPLIT>'],
}
FFN_EXPECT = {
    1024: ['gemm_f32_kernel<64,64,32,4w,FAST,SPLIT>', 'gemm_small_kernel<8w,FAST>', 'gemm_f32_kernel<64,64,32,4w,FAST,SPLIT>', 'gemm_small_pair_kernel<4w,FAST>', 'gemm_small_kernel<8w,FAST>'],
    2048: ['gemm_f32_kernel<128,64,32,4w,FAST,SPLIT>', 'gemm_small_kernel<4w,FAST>', 'gemm_f32_kernel<128,64,32,4w,FAST,SPLIT>', 'gemm_small_pair_kernel<4w,FAST>', 'gemm_small_kernel<4w,FAST>'],
}


@pytest.mark.parametrize('M,K,N,what', LINEAR_CASES, ids=lambda v: str(v).replace(' ', '_')[:40])
def test_linear_layers_at_bench_shapes_vs_float64(dev, M, K, N, what):
    """y = relu(x W^T + b), dX, dW, db of hip.linear at the row / width combinations of the headline step."""
    from layoutdetr_amd.hip import core
    from layoutdetr_amd.hip.linear import linear
    torch.manual_seed(400 + M + N)
    x = torch.randn(M, K); w = torch.randn(N, K) / math.sqrt(K); b = torch.randn(N) * 0.1 + 6.0       # (+6 sigma: the ReLU stays on one branch, see above)
    g = torch.randn(M, N)
    xr, wr, br = [t.double().requires_grad_(True) for t in (x, w, b)]
    yr = torch.relu(xr @ wr.t() + br)
    yr.backward(g.double())
    xd, wd, bd = [t.to(dev).requires_grad_(True) for t in (x, w, b)]
    with Launches() as L:
        y = linear(xd, wd, bd, act=core.ACT_RELU)
        y.backward(g.to(dev))
    torch.cuda.synchronize()
    expect(L.seen, {'gemm': LINEAR_EXPECT[(M, K, N)]}, f'{what} [{M} x {K}] -> {N}')
    errs = dict(y=rel(y, yr), dx=rel(xd.grad, xr.grad), dw=rel(wd.grad, wr.grad), db=rel(bd.grad, br.grad))
    print('   ', {k: f'{v:.1e}' for k, v in errs.items()})
    for k, v in errs.items():
        assert v <= 2e-5, f'{what}: {k} is {v:.2e} from float64'
    # the same layer with its parameters re-homed into a flat gradient buffer (training_loop.FlatModule): dX + dW as ONE C-ABI call
    # (ldetr_gemm_pair_f32), the weight / bias gradients accumulated in place on top of what the buffer already holds
    wp, bp = torch.nn.Parameter(w.to(dev)), torch.nn.Parameter(b.to(dev))
    g0w, g0b = torch.randn(N, K, device=dev), torch.randn(N, device=dev)
    wp.grad, bp.grad = g0w.clone(), g0b.clone()
    wp._ldetr_flat = bp._ldetr_flat = True
    xd2 = x.to(dev).requires_grad_(True)
    with Launches() as L2:
        linear(xd2, wp, bp, act=core.ACT_RELU).backward(g.to(dev))
    torch.cuda.synchronize()
    expect(L2.seen, None, f'{what}, flat gradient buffer')
    e2 = dict(dx=rel(xd2.grad, xr.grad), dw=rel(wp.grad - g0w, wr.grad), db=rel(bp.grad - g0b, br.grad))
    for k, v in e2.items():
        assert v <= 2e-5, f'{what} (flat gradients): {k} is {v:.2e} from float64'


@pytest.mark.parametrize('M', [1024, 2048])
def test_large_feed_forward_node_at_bench_shapes_vs_float64(dev, M):
    """hip.ffn.ffn_large (the 64-token encoders' feed-forward block as one autograd node: hidden gradient out of the first GEMM's epilogue, paired weight
    gradients) at 16 x 64 and 32 x 64 rows against float64, dropout off."""
    from layoutdetr_amd.hip import ffn as hffn
    torch.manual_seed(500 + M)
    l1 = torch.nn.Linear(256, 2048); l2 = torch.nn.Linear(2048, 256)
    with torch.no_grad():
        l1.bias.add_(4.0)
    x = torch.randn(M, 256); g = torch.randn(M, 256)
    xr = x.double().requires_grad_(True)
    p64 = [p.detach().double().requires_grad_(True) for p in (l1.weight, l1.bias, l2.weight, l2.bias)]
    yr = torch.relu(xr @ p64[0].t() + p64[1]) @ p64[2].t() + p64[3]
    yr.backward(g.double())
    l1.to(dev); l2.to(dev)
    xd = x.to(dev).requires_grad_(True)
    assert hffn.large_usable(xd, l1, l2)
    with Launches() as L:
        out = hffn.ffn_large(xd, l1, l2, 0.0)
        y = out[0] if isinstance(out, tuple) else out
        y.backward(g.to(dev))
    torch.cuda.synchronize()
    expect(L.seen, {'gemm': FFN_EXPECT[M]}, f'feed-forward node, {M} rows')
    errs = dict(y=rel(y, yr), dx=rel(xd.grad, xr.grad), dw1=rel(l1.weight.grad, p64[0].grad), db1=rel(l1.bias.grad, p64[1].grad), dw2=rel(l2.weight.grad, p64[2].grad),
                db2=rel(l2.bias.grad, p64[3].grad))
    print('   ', {k: f'{v:.1e}' for k, v in errs.items()})
    for k, v in errs.items():
        assert v <= 2e-5, f'feed-forward node at {M} rows: {k} is {v:.2e} from float64'
