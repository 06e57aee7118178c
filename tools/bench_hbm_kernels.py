"""Stand-alone rates of the HBM-bound kernels at the StyleGAN2 Decoder's 256x256 / 128x128 shapes (B=16): upfirdn2d (up-layer FIR),
bias_act forward / backward.  Algorithmic bytes per SURVEY 8(d): FIR 8 B per output element, bias_act 8 (fwd) / 12 (bwd).
hipGraph-replayed (no host gaps), HIP events around the replay.   python tools/bench_hbm_kernels.py [short]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from layoutdetr_amd.torch_utils.ops import bias_act, upfirdn2d


def timed(fn, reps=20, prepare=None):
    """prepare(): builds fn on the capture stream (autograd graphs must be created on the stream their backward is captured on)."""
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        if prepare is not None:
            fn = prepare()
        fn(); fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


def measure(B=16):
    """-> list of dict(kernel, shape, bytes, us, tbps)."""
    dev = torch.device('cuda')
    f = upfirdn2d.setup_filter([1, 3, 3, 1], device=dev)
    out = []
    for res, C in ((256, 32), (128, 64), (64, 128)):
        x = torch.randn(B, C, res + 1, res + 1, device=dev).contiguous(memory_format=torch.channels_last)
        y = [None]
        t = timed(lambda: y.__setitem__(0, upfirdn2d._kernel_call(x, f, 1, 1, 1, 1, 1, 1, 1, 1, False, 4.0)))
        nbytes = 4 * (x.numel() + y[0].numel())
        out.append(dict(kernel='upfirdn2d 4x4 (up-layer FIR)', shape=f'{B}x{C}x{res}x{res}', bytes=nbytes, us=t * 1e6, tbps=nbytes / t / 1e12))
        xa = torch.randn(B, C, res, res, device=dev).contiguous(memory_format=torch.channels_last)
        b = torch.randn(C, device=dev)
        t = timed(lambda: bias_act.bias_act(xa, b, act='lrelu'))
        nbytes = 8 * xa.numel()
        out.append(dict(kernel='bias_act lrelu fwd', shape=f'{B}x{C}x{res}x{res}', bytes=nbytes, us=t * 1e6, tbps=nbytes / t / 1e12))
        def prep():
            xr = xa.clone().requires_grad_(True)
            yb = bias_act.bias_act(xr, b, act='lrelu')
            dy = torch.randn_like(yb)
            return lambda: torch.autograd.grad(yb, xr, dy, retain_graph=True)
        t = timed(None, prepare=prep)
        nbytes = 12 * xa.numel()
        out.append(dict(kernel='bias_act lrelu bwd', shape=f'{B}x{C}x{res}x{res}', bytes=nbytes, us=t * 1e6, tbps=nbytes / t / 1e12))
    return out


def measure_named(B=16):
    """The fractions north_star names, each as its own roofline entry: HBM fraction of the HBM-bound kernels (FIR, bias_act, Adam, EMA,
    the fused modulated-conv layer at 256x256) and MFMA utilisation of DETR cross-attention, in-kernel and device-level."""
    import math
    from layoutdetr_amd.hip import attention as hattn
    from layoutdetr_amd.hip import core, modconv
    dev = torch.device('cuda')
    HBM, MFMA = 8.0, 157.3
    out = []
    # the practical ceiling of THIS box: a plain device copy (read + write) at a footprint far beyond the 256 MB Infinity Cache and at the FIR's footprint
    copy_rate = {}
    for nm, elems in (('1.44 GB (the optimiser\'s footprint at 90 M parameters)', 180 << 20), ('270 MB (the 256 x 256 FIR\'s footprint)', 34 << 20)):
        a_ = torch.empty(elems, device=dev); b_ = torch.empty(elems, device=dev)
        t = timed(lambda: b_.copy_(a_), reps=5)
        copy_rate[nm] = 8.0 * elems / t / 1e12
        out.append(dict(kernel='device copy (ceiling probe: aten copy, read + write)', shape=nm, bound='hbm', achieved=round(copy_rate[nm], 3), peak=HBM, unit='TB/s', frac=round(copy_rate[nm] / HBM, 4),
                        algorithmic_bytes=8 * elems, us=round(t * 1e6, 1), note='what a pure stream reaches on this box; the fractions below should be read against it as well as against 8 TB/s'))
        del a_, b_
    ceiling = min(copy_rate.values())
    for r in measure(B):
        out.append(dict(kernel=r['kernel'], shape=r['shape'], bound='hbm', achieved=round(r['tbps'], 3), peak=HBM, unit='TB/s', frac=round(r['tbps'] / HBM, 4),
                        frac_of_copy_rate=round(r['tbps'] / ceiling, 3), algorithmic_bytes=r['bytes'], us=round(r['us'], 1)))
    # fused modulated 3x3 conv layer at 256x256, 32 -> 32 channels (styles in the loader, demodulation + bias + lrelu in the epilogue): ONE launch;
    # algorithmic bytes 4 (Cin r^2 + Cout r^2) per sample; it is MFMA-bound (N = 32), so both fractions are given
    C, R = 32, 256
    x = torch.randn(B, R, R, C, device=dev); w = torch.randn(C, C, 3, 3, device=dev).contiguous(memory_format=torch.channels_last)
    st = torch.rand(B, C, device=dev) + 0.5; bias = torch.randn(C, device=dev)
    with torch.no_grad():
        t = timed(lambda: modconv.modconv3x3(x, w, st, bias))
    nbytes, fl = 4 * 2 * B * R * R * C, 2.0 * B * R * R * C * C * 9
    out.append(dict(kernel='modulated conv3x3 layer fwd (1 launch + demod coefficients)', shape=f'{B}x{C}->{C}x{R}x{R}', bound='mfma', achieved=round(fl / t / 1e12, 2), peak=MFMA,
                    unit='TFLOP/s', frac=round(fl / t / 1e12 / MFMA, 4), algorithmic_bytes=nbytes, hbm_tbps=round(nbytes / t / 1e12, 3), hbm_frac=round(nbytes / t / 1e12 / HBM, 4),
                    us=round(t * 1e6, 1), note='N = 32 output channels: MFMA-bound (123 us at the f32 matrix peak vs 34 us at 8 TB/s); hbm_frac is what the north_star target reads'))
    # optimiser: fused sanitize + Adam over (p, g, m, v) = 28 B per parameter; EMA lerp = 12 B per parameter, at D's parameter count
    P = 89928156
    pbuf, g, m, vv, pe = (torch.randn(P, device=dev) * 0.01 for _ in range(5))
    vv.abs_()
    L = core.lib()
    t = timed(lambda: core.check(L.ldetr_adam_step_f32(core.ptr(pbuf), core.ptr(g), core.ptr(m), core.ptr(vv), P, 3, 1e-5, 0.0, 0.99, 1e-8, 1, 1.0, 0.0, 1e5, -1e5, core.stream())), reps=5)
    out.append(dict(kernel='adam_kernel (+ /world + nan_to_num)', shape=f'{P} params', bound='hbm', achieved=round(28.0 * P / t / 1e12, 3), peak=HBM, unit='TB/s', frac=round(28.0 * P / t / 1e12 / HBM, 4),
                    frac_of_copy_rate=round(28.0 * P / t / 1e12 / ceiling, 3), algorithmic_bytes=28 * P, us=round(t * 1e6, 1)))
    t = timed(lambda: core.check(L.ldetr_ema_lerp_f32(core.ptr(pe), core.ptr(pbuf), P, 0.999, core.stream())), reps=5)
    out.append(dict(kernel='ema_kernel', shape=f'{P} params', bound='hbm', achieved=round(12.0 * P / t / 1e12, 3), peak=HBM, unit='TB/s', frac=round(12.0 * P / t / 1e12 / HBM, 4),
                    frac_of_copy_rate=round(12.0 * P / t / 1e12 / ceiling, 3), algorithmic_bytes=12 * P, us=round(t * 1e6, 1)))
    del pbuf, g, m, vv, pe
    # DETR cross-attention (decoder layer, detr_transformer.py:277-280): (8B, Lq=9, S=64, dh=32)
    H, Lq, S, dh = 8, 9, 64, 32
    q = torch.randn(B * Lq, H * dh, device=dev); k = torch.randn(B * S, H * dh, device=dev); v = torch.randn(B * S, H * dh, device=dev)
    with torch.no_grad():
        t = timed(lambda: hattn.attention(q, k, v, None, B, H, Lq, S, 0.0))
    fl = 4.0 * Lq * S * dh * B * H
    # in-kernel: one wave per (b, h, 16-query tile) issues (S/16)(dh/4) MFMAs for K Q^T and as many for P V, v_mfma_f32_16x16x4_f32 at 32 cycles per SIMD issue
    mfma_cycles = 2 * (S // 16) * (dh // 4) * 32
    out.append(dict(kernel='DETR cross-attention fwd (attn_fwd_kernel)', shape=f'(b*h={B * H}, Lq={Lq}, Lk={S}, dh={dh})', bound='latency', achieved=round(fl / t / 1e12, 4), peak=MFMA, unit='TFLOP/s',
                    frac=round(fl / t / 1e12 / MFMA, 6), us=round(t * 1e6, 2), mfma_util_device=round(fl / t / 1e12 / MFMA, 6),
                    mfma_util_in_kernel=round(mfma_cycles / (t * 2.4e9), 4),
                    note='1.2 MFLOP per sample and layer on 128 waves: launch / latency-bound at any utilisation (SURVEY 7); in-kernel = MFMA issue cycles of a wave / kernel duration at 2.4 GHz'))
    # the same cross-attention as the training step runs it since round 3: query projection + attention + per-head output projection of the
    # decoder layer's sub-block in ONE launch (csrc/mha_small.hip: mha_cross_fwd_kernel), and the self-attention sub-block likewise
    d = H * dh
    Wi = torch.randn(3 * d, d, device=dev) * 0.05; bi = torch.randn(3 * d, device=dev); Wo = torch.randn(d, d, device=dev) * 0.05
    x = torch.randn(B * Lq, d, device=dev)
    qb = torch.empty(B * Lq, d, device=dev); ob = torch.empty(B * Lq, d, device=dev); lse = torch.empty(B * H * Lq, device=dev); yp = torch.empty(H, B * Lq, d, device=dev)
    qkv = torch.empty(B * Lq, 3 * d, device=dev)
    L = core.lib()
    t = timed(lambda: core.check(L.ldetr_mha_cross_fwd_f32(core.ptr(x), d, core.ptr(Wi), core.ptr(bi), core.ptr(k), d, core.ptr(v), d, core.ptr(Wo), None, core.ptr(qb), core.ptr(ob),
                                                           core.ptr(lse), core.ptr(yp), B, Lq, S, d, H, 1.0 / dh ** 0.5, 0.0, 0, None, core.stream())))
    fl2 = 2.0 * B * Lq * d * d * 2 + fl
    cyc = (128 // 4 + 2 * (S // 16) * (dh // 4) + 128 // 4) * 32          # per-SIMD MFMA issue cycles of a (sample, head) block: projection, attention (one wave), output projection
    out.append(dict(kernel='DETR cross-attention sub-block fwd (mha_cross_fwd_kernel: q projection + attention + out projection)', shape=f'(b*h={B * H}, Lq={Lq}, Lk={S}, dh={dh})', bound='latency',
                    achieved=round(fl2 / t / 1e12, 4), peak=MFMA, unit='TFLOP/s', frac=round(fl2 / t / 1e12 / MFMA, 6), us=round(t * 1e6, 2), mfma_util_device=round(fl2 / t / 1e12 / MFMA, 6),
                    mfma_util_in_kernel=round(cyc / (t * 2.4e9), 4), note='replaces three launches (5.5 + 8.9 + 5.5 us stand-alone); rows 10..15 of every 16-row MFMA tile are padding'))
    t = timed(lambda: core.check(L.ldetr_mha_small_fwd_f32(core.ptr(x), d, core.ptr(Wi), core.ptr(bi), core.ptr(Wo), None, core.ptr(qkv), core.ptr(ob), core.ptr(lse), core.ptr(yp),
                                                           B, Lq, d, H, 1.0 / dh ** 0.5, 0.0, 0, None, core.stream())))
    fl3 = 2.0 * B * Lq * d * (3 * d) + 4.0 * Lq * Lq * dh * B * H + 2.0 * B * Lq * d * d
    out.append(dict(kernel='self-attention sub-block fwd (mha_small_fwd_kernel: packed projection + attention + out projection)', shape=f'(b*h={B * H}, L={Lq}, dh={dh})', bound='latency',
                    achieved=round(fl3 / t / 1e12, 4), peak=MFMA, unit='TFLOP/s', frac=round(fl3 / t / 1e12 / MFMA, 6), us=round(t * 1e6, 2),
                    mfma_util_in_kernel=round((96 + 16 + 32) * 32 / (t * 2.4e9), 4), note='replaces three launches (8 + 6 + 6 us)'))
    return out


if __name__ == '__main__':
    for r in measure():
        print(f"{r['kernel']:32s} {r['shape']:18s} {r['bytes'] / 1e6:8.1f} MB {r['us']:8.1f} us {r['tbps']:6.2f} TB/s ({r['tbps'] / 8.0:.2f} of 8 TB/s)")
    import json
    for r in ([] if 'short' in sys.argv[1:] else measure_named()):
        print(json.dumps(r))
