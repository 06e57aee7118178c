"""Stand-alone rates of the HBM-bound kernels at the StyleGAN2 Decoder's 256x256 / 128x128 shapes (B=16): upfirdn2d (up-layer FIR),
bias_act forward / backward.  Algorithmic bytes per SURVEY 8(d): FIR 8 B per output element, bias_act 8 (fwd) / 12 (bwd).
hipGraph-replayed (no host gaps), HIP events around the replay.   python tools/bench_hbm_kernels.py"""
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


if __name__ == '__main__':
    for r in measure():
        print(f"{r['kernel']:32s} {r['shape']:18s} {r['bytes'] / 1e6:8.1f} MB {r['us']:8.1f} us {r['tbps']:6.2f} TB/s ({r['tbps'] / 8.0:.2f} of 8 TB/s)")
