"""Development driver of the plane-format engine (csrc/p3_engine.hip): probes, parity against fp64 torch, timings.
Usage: python tools/p3_dev.py [check] [checkb] [bench] [benchb | benchd | benchw] [pair [graph]]"""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from layoutdetr_amd.hip import core
from layoutdetr_amd import _lib
dev = torch.device('cuda:0')
L = core.lib()


def p3_split(x2d):
    rows, C = x2d.shape
    out = torch.empty(rows * C * 6, dtype=torch.uint8, device=dev)
    core.check(L.ldetr_p3_split_f32(core.ptr(x2d), x2d.stride(0), core.ptr(out), rows, C, core.stream()), 'split')
    return out


def p3_merge(p, rows, C):
    out = torch.empty(rows, C, dtype=torch.float32, device=dev)
    core.check(L.ldetr_p3_merge_f32(core.ptr(p), core.ptr(out), C, rows, C, core.stream()), 'merge')
    return out


def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); g.replay(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / (2 * n) * 1e-3


def conv_ref(x, w, stride, pad):
    return F.conv2d(x.permute(0, 3, 1, 2).double(), w.permute(0, 3, 1, 2).double(), stride=stride, padding=pad).permute(0, 2, 3, 1)


def run_fwd(xp, N, H, Ci, wp, Co, k, s, pad, ep, yp, yf):
    return L.ldetr_p3_conv2d_fwd(core.ptr(xp), N, H, H, Ci, core.ptr(wp), Co, k, k, s, pad, ctypes.byref(ep) if ep is not None else None,
                                 core.ptr(yp), core.ptr(yf), core.stream())


def check_case(N, H, Ci, Co, k, s, pad, full_ep=True):
    torch.manual_seed(1)
    x = torch.randn(N, H, H, Ci, device=dev) * torch.exp(torch.randn(N, H, H, Ci, device=dev))
    w = torch.randn(Co, k, k, Ci, device=dev) / (k * Ci ** 0.5)
    OH = (H + 2 * pad - k) // s + 1
    xp = p3_split(x.reshape(-1, Ci)); wp = p3_split(w.reshape(Co, -1))
    assert torch.equal(p3_merge(xp, N * H * H, Ci), x.reshape(-1, Ci)), 'split/merge is not exact'
    yf = torch.empty(N, OH, OH, Co, device=dev); yp = torch.empty(N * OH * OH * Co * 6, dtype=torch.uint8, device=dev)
    ep = None
    ref = conv_ref(x, w, s, pad)
    if full_ep:
        sc = torch.rand(Co, device=dev) + 0.5; sh = torch.randn(Co, device=dev); res = torch.randn(N, OH, OH, Co, device=dev)
        resp = p3_split(res.reshape(-1, Co))
        ep = _lib.P3Epilogue(); ep.alpha = 1.0; ep.col_scale = sc.data_ptr(); ep.col_bias = sh.data_ptr(); ep.residual_p3 = resp.data_ptr(); ep.relu = 1
        ref = torch.relu(ref * sc.double() + sh.double() + res.double())
    core.check(run_fwd(xp, N, H, Ci, wp, Co, k, s, pad, ep, yp, yf), 'fwd')
    torch.cuda.synchronize()
    err = (yf.double() - ref).abs().max().item() / ref.abs().max().item()
    rms = ((yf.double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
    same = torch.equal(p3_merge(yp, N * OH * OH, Co), yf.reshape(-1, Co))
    print(f'check N={N} H={H} {Ci}->{Co} k{k} s{s} ep={full_ep}: max err/max {err:.2e} rms {rms:.2e} p3==f32 {same}', flush=True)
    assert err < 2e-6 and same


def p3_weight_bwd(w, scale=None):
    Co, k, _, Ci = w.shape
    out = torch.empty(Ci * k * k * Co * 6, dtype=torch.uint8, device=dev)
    core.check(L.ldetr_p3_weight_bwd(core.ptr(w), core.ptr(scale), core.ptr(out), Co, k, k, Ci, core.stream()), 'weight_bwd')
    return out


def check_bwd(N, H, Ci, Co, k, s, pad):
    torch.manual_seed(2)
    OH = (H + 2 * pad - k) // s + 1
    x = torch.randn(N, H, H, Ci, device=dev)
    w = torch.randn(Co, k, k, Ci, device=dev) / (k * Ci ** 0.5)
    dy = torch.randn(N, OH, OH, Co, device=dev) * torch.exp(torch.randn(N, OH, OH, Co, device=dev))
    sc = torch.rand(Co, device=dev) + 0.5
    xd = x.double().permute(0, 3, 1, 2).requires_grad_(True); wd = w.double().permute(0, 3, 1, 2).requires_grad_(True)
    y = F.conv2d(xd, wd, stride=s, padding=pad)
    gx, gw = torch.autograd.grad(y, (xd, wd), (dy.double() * sc.double()).permute(0, 3, 1, 2))
    gx = gx.permute(0, 2, 3, 1); gw = gw.permute(0, 2, 3, 1)
    # data gradient with mask (x > 0) and a residual
    res = torch.randn(N, H, H, Ci, device=dev)
    dyp = p3_split(dy.reshape(-1, Co)); xp = p3_split(x.reshape(-1, Ci)); resp = p3_split(res.reshape(-1, Ci)); wb = p3_weight_bwd(w, sc)
    ep = _lib.P3Epilogue(); ep.alpha = 1.0; ep.residual_p3 = resp.data_ptr(); ep.relu_mask_p3 = xp.data_ptr()
    dxf = torch.empty(N, H, H, Ci, device=dev); dxp = torch.empty(N * H * H * Ci * 6, dtype=torch.uint8, device=dev)
    core.check(L.ldetr_p3_conv2d_bwd_data(core.ptr(dyp), N, OH, OH, Co, core.ptr(wb), Ci, k, k, s, pad, H, H, ctypes.byref(ep), core.ptr(dxp), core.ptr(dxf), core.stream()), 'bwd_data')
    ref = torch.where(x > 0, gx + res.double(), torch.zeros_like(gx))
    e1 = (dxf.double() - ref).abs().max().item() / ref.abs().max().item()
    same = torch.equal(p3_merge(dxp, N * H * H, Ci), dxf.reshape(-1, Ci))
    # weight gradient (accumulating onto an existing buffer)
    dw0 = torch.randn(Co, k, k, Ci, device=dev); dw = dw0.clone()
    core.check(L.ldetr_p3_conv2d_bwd_weight(core.ptr(xp), N, H, H, Ci, core.ptr(dyp), Co, k, k, s, pad, core.ptr(sc), core.ptr(dw), core.stream()), 'bwd_weight')
    torch.cuda.synchronize()
    e2 = ((dw - dw0).double() - gw).abs().max().item() / gw.abs().max().item()
    print(f'check bwd N={N} H={H} {Ci}->{Co} k{k} s{s}: dx err {e1:.2e} p3==f32 {same} | dw err {e2:.2e}', flush=True)
    assert e1 < 3e-6 and same and e2 < 3e-6


P3_ONLY = bool(os.environ.get('P3_ONLY'))


def bench_bwd(name, N, H, Ci, Co, k, s, pad, what='dw'):
    OH = (H + 2 * pad - k) // s + 1
    x = torch.randn(N, H, H, Ci, device=dev); w = torch.randn(Co, k, k, Ci, device=dev) / (k * Ci ** 0.5); dy = torch.randn(N, OH, OH, Co, device=dev)
    sc = torch.rand(Co, device=dev) + 0.5
    dyp = p3_split(dy.reshape(-1, Co)); xp = p3_split(x.reshape(-1, Ci)); wb = p3_weight_bwd(w, sc)
    ep = _lib.P3Epilogue(); ep.alpha = 1.0; ep.relu_mask_p3 = xp.data_ptr()
    dxp = torch.empty(N * H * H * Ci * 6, dtype=torch.uint8, device=dev); dw = torch.zeros(Co, k, k, Ci, device=dev)
    fl = 2.0 * N * OH * OH * Co * k * k * Ci
    t1 = timeit(lambda: L.ldetr_p3_conv2d_bwd_data(core.ptr(dyp), N, OH, OH, Co, core.ptr(wb), Ci, k, k, s, pad, H, H, ctypes.byref(ep), core.ptr(dxp), None, core.stream())) if 'd' in what else 1.0
    t2 = timeit(lambda: L.ldetr_p3_conv2d_bwd_weight(core.ptr(xp), N, H, H, Ci, core.ptr(dyp), Co, k, k, s, pad, core.ptr(sc), core.ptr(dw), core.stream())) if 'w' in what else 1.0
    if P3_ONLY:
        print(f'{name:24s} bwdD p3 {t1*1e6:7.1f}us {fl/t1/1e12:6.1f}TF | bwdW p3 {t2*1e6:7.1f}us {fl/t2/1e12:6.1f}TF', flush=True)
        return
    dx = torch.empty_like(x); dyt = core.tensor4_nhwc(dy); xt = core.tensor4_nhwc(x)
    epb = core.epilogue(mask_src=x.reshape(-1, Ci), mask_mode=1)
    t3 = timeit(lambda: L.ldetr_conv2d_bwd_data_f32(core.ptr(dy), ctypes.byref(dyt), core.ptr(w), Ci, k, k, s, pad, core.ptr(dx), Ci, H, H, core.ptr(sc), 0, ctypes.byref(epb), core.stream()))
    t4 = timeit(lambda: L.ldetr_conv2d_bwd_weight_f32(core.ptr(x), ctypes.byref(xt), core.ptr(dy), ctypes.byref(dyt), core.ptr(dw), k, k, s, pad, 0, None, 0, core.ptr(sc), 0, 1, core.stream()))
    print(f'{name:24s} bwdD p3 {t1*1e6:7.1f}us {fl/t1/1e12:6.1f}TF (f32 eng {t3*1e6:7.1f}us {fl/t3/1e12:6.1f}) | bwdW p3 {t2*1e6:7.1f}us {fl/t2/1e12:6.1f}TF (f32 eng {t4*1e6:7.1f}us {fl/t4/1e12:6.1f})', flush=True)


def pair_test(name, N, H, Ci, Co, k, s, pad, reps=30):
    """Do the data gradient and the weight gradient of one layer overlap when they may?  Serial on one stream vs the two on two streams."""
    OH = (H + 2 * pad - k) // s + 1
    x = torch.randn(N, H, H, Ci, device=dev); w = torch.randn(Co, k, k, Ci, device=dev) / (k * Ci ** 0.5); dy = torch.randn(N, OH, OH, Co, device=dev)
    sc = torch.rand(Co, device=dev) + 0.5
    dyp = p3_split(dy.reshape(-1, Co)); xp = p3_split(x.reshape(-1, Ci)); wb = p3_weight_bwd(w, sc)
    ep = _lib.P3Epilogue(); ep.alpha = 1.0; ep.relu_mask_p3 = xp.data_ptr()
    dxp = torch.empty(N * H * H * Ci * 6, dtype=torch.uint8, device=dev); dw = torch.zeros(Co, k, k, Ci, device=dev)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    def dgrad(st): return L.ldetr_p3_conv2d_bwd_data(core.ptr(dyp), N, OH, OH, Co, core.ptr(wb), Ci, k, k, s, pad, H, H, ctypes.byref(ep), core.ptr(dxp), None, ctypes.c_void_p(st.cuda_stream))
    def wgrad(st): return L.ldetr_p3_conv2d_bwd_weight(core.ptr(xp), N, H, H, Ci, core.ptr(dyp), Co, k, k, s, pad, core.ptr(sc), core.ptr(dw), ctypes.c_void_p(st.cuda_stream))
    def run(mode):
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e2 = torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(2_000_000)            # let the host run ahead
        s1.wait_stream(torch.cuda.current_stream()); s2.wait_stream(torch.cuda.current_stream())
        e0.record(s1)
        s2.wait_event(e0)
        for _ in range(reps):
            if mode == 'serial':
                dgrad(s1); wgrad(s1)
            elif mode == 'd':
                dgrad(s1)
            elif mode == 'w':
                wgrad(s1)
            else:
                dgrad(s1); wgrad(s2)
        e2.record(s2); s1.wait_event(e2); e1.record(s1)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    def graphed(par):
        g = torch.cuda.CUDAGraph()
        cs = torch.cuda.Stream()
        with torch.cuda.graph(g, stream=cs):
            for _ in range(reps):
                if par:
                    s2.wait_stream(cs)
                    dgrad(cs); wgrad(s2)
                    cs.wait_stream(s2)
                else:
                    dgrad(cs); wgrad(cs)
        g.replay(); torch.cuda.synchronize()
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / reps * 1e3
    run('serial')
    if 'graph2' in sys.argv:      # two GRAPHS (one per kernel kind) replayed on two streams: do whole graphs overlap?
        def cap(fn):
            g = torch.cuda.CUDAGraph(); cs = torch.cuda.Stream()
            with torch.cuda.graph(g, stream=cs):
                for _ in range(reps): fn(cs)
            return g
        gd, gw = cap(dgrad), cap(wgrad)
        def rep2(par):
            torch.cuda.synchronize()
            a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
            cur = torch.cuda.current_stream()
            a.record()
            if par:
                s2.wait_stream(cur)
                gd.replay()
                with torch.cuda.stream(s2): gw.replay()
                cur.wait_stream(s2)
            else:
                gd.replay(); gw.replay()
            b.record(); torch.cuda.synchronize()
            return a.elapsed_time(b) / reps * 1e3
        rep2(False); rep2(True)
        ts, tp = rep2(False), rep2(True)
        print(f'{name:24s} two graphs back to back {ts:6.1f}us  on two streams {tp:6.1f}us ({tp / ts:.2f}x)', flush=True)
        return
    if 'graph' in sys.argv:
        gs, gp = graphed(False), graphed(True)
        print(f'{name:24s} graph serial {gs:6.1f}us  graph fork/join per layer {gp:6.1f}us ({gp / gs:.2f}x)', flush=True)
        return
    def fused(st): return L.ldetr_p3_conv2d_bwd_pair(core.ptr(dyp), N, OH, OH, Co, core.ptr(wb), core.ptr(xp), Ci, k, k, s, pad, H, H, ctypes.byref(ep), core.ptr(dxp), None,
                                                     core.ptr(sc), core.ptr(dw), None, ctypes.c_void_p(st.cuda_stream))
    def run_fused():
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(2_000_000)
        s1.wait_stream(torch.cuda.current_stream())
        e0.record(s1)
        for _ in range(reps):
            fused(s1)
        e1.record(s1)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    td, tw, ts, tp = run('d'), run('w'), run('serial'), run('par')
    run_fused(); tf = run_fused()
    print(f'{name:24s} dgrad {td:6.1f}us wgrad {tw:6.1f}us  serial {ts:6.1f}us  two streams {tp:6.1f}us ({tp / ts:.2f}x)  one launch {tf:6.1f}us ({tf / ts:.2f}x)', flush=True)


def bench_case(name, N, H, Ci, Co, k, s, pad):
    x = torch.randn(N, H, H, Ci, device=dev); w = torch.randn(Co, k, k, Ci, device=dev) / (k * Ci ** 0.5)
    OH = (H + 2 * pad - k) // s + 1
    xp = p3_split(x.reshape(-1, Ci)); wp = p3_split(w.reshape(Co, -1))
    yp = torch.empty(N * OH * OH * Co * 6, dtype=torch.uint8, device=dev)
    sc = torch.rand(Co, device=dev) + 0.5; sh = torch.randn(Co, device=dev); res = torch.randn(N, OH, OH, Co, device=dev)
    resp = p3_split(res.reshape(-1, Co))
    ep = _lib.P3Epilogue(); ep.alpha = 1.0; ep.col_scale = sc.data_ptr(); ep.col_bias = sh.data_ptr(); ep.residual_p3 = resp.data_ptr(); ep.relu = 1
    fl = 2.0 * N * OH * OH * Co * k * k * Ci
    t = timeit(lambda: run_fwd(xp, N, H, Ci, wp, Co, k, s, pad, ep, yp, None))
    if P3_ONLY:
        print(f'{name:24s} M={N*OH*OH:6d} N={Co:4d} K={k*k*Ci:5d}  p3 {t*1e6:7.1f}us {fl/t/1e12:6.1f}TF', flush=True)
        return
    # the f32 engine on the same problem
    y = torch.empty(N, OH, OH, Co, device=dev); xt = core.tensor4_nhwc(x)
    ep0 = core.epilogue(col_scale=sc, col_bias=sh, residual=res.reshape(-1, Co), act=core.ACT_RELU)
    t0 = timeit(lambda: L.ldetr_conv2d_fwd_f32(core.ptr(x), ctypes.byref(xt), core.ptr(w), Co, k, k, s, pad, core.ptr(y), Co, OH, OH, None, 0, ctypes.byref(ep0), core.stream()))
    print(f'{name:24s} M={N*OH*OH:6d} N={Co:4d} K={k*k*Ci:5d}  p3 {t*1e6:7.1f}us {fl/t/1e12:6.1f}TF | f32 engine {t0*1e6:7.1f}us {fl/t0/1e12:6.1f}TF', flush=True)


CASES = [('l1 1x1 64->64', 64, 64, 64, 1, 1, 0), ('l1 3x3 64->64', 64, 64, 64, 3, 1, 1), ('l1 1x1 64->256', 64, 64, 256, 1, 1, 0), ('l1 1x1 256->64', 64, 256, 64, 1, 1, 0),
         ('l2 1x1 256->128', 64, 256, 128, 1, 1, 0), ('l2 3x3 128->128 s2', 64, 128, 128, 3, 2, 1), ('l2 1x1 128->512', 32, 128, 512, 1, 1, 0), ('l2 1x1 256->512 s2', 64, 256, 512, 1, 2, 0),
         ('l2 1x1 512->128', 32, 512, 128, 1, 1, 0), ('l2 3x3 128->128', 32, 128, 128, 3, 1, 1),
         ('l3 3x3 256->256 s2', 32, 256, 256, 3, 2, 1), ('l3 1x1 256->1024', 16, 256, 1024, 1, 1, 0), ('l3 1x1 1024->256', 16, 1024, 256, 1, 1, 0), ('l3 3x3 256->256', 16, 256, 256, 3, 1, 1),
         ('l4 3x3 512->512 s2', 16, 512, 512, 3, 2, 1), ('l4 1x1 512->2048', 8, 512, 2048, 1, 1, 0), ('l4 1x1 2048->512', 8, 2048, 512, 1, 1, 0), ('l4 3x3 512->512', 8, 512, 512, 3, 1, 1),
         ('l3 1x1 512->256', 32, 512, 256, 1, 1, 0), ('l3 1x1 512->1024 s2', 32, 512, 1024, 1, 2, 0), ('l4 1x1 1024->512', 16, 1024, 512, 1, 1, 0), ('l4 1x1 1024->2048 s2', 16, 1024, 2048, 1, 2, 0),
         ('sg 3x3 512->512 @16', 16, 512, 512, 3, 1, 1), ('sg 3x3 128->128 @64', 64, 128, 128, 3, 1, 1)]

def main():
    what = sys.argv[1:] or ['check', 'checkb', 'bench', 'benchb']
    if 'check' in what:
        check_case(1, 8, 32, 64, 1, 1, 0, False)
        check_case(2, 8, 64, 64, 3, 1, 1, False)
        check_case(2, 9, 64, 72, 3, 1, 1, True)
        check_case(2, 16, 64, 128, 3, 2, 1, True)
        check_case(3, 16, 128, 256, 1, 2, 0, True)
        check_case(4, 32, 128, 128, 3, 1, 1, True)
        check_case(16, 8, 512, 512, 3, 1, 1, True)     # split-K
        check_case(16, 16, 1024, 256, 1, 1, 0, True)   # split-K
    if 'checkb' in what:
        check_bwd(1, 8, 32, 32, 1, 1, 0)
        check_bwd(2, 8, 64, 64, 3, 1, 1)
        check_bwd(2, 9, 64, 96, 3, 1, 1)
        check_bwd(2, 16, 64, 128, 3, 2, 1)
        check_bwd(3, 16, 128, 256, 1, 2, 0)
        check_bwd(4, 32, 128, 128, 3, 1, 1)
        check_bwd(16, 8, 512, 512, 3, 1, 1)
        check_bwd(16, 16, 1024, 256, 1, 1, 0)
    B = int(os.environ.get('B', '16'))
    if 'bench' in what:
        for c in CASES:
            bench_case(c[0], B, *c[1:])
    if 'benchb' in what:
        for c in CASES:
            bench_bwd(c[0], B, *c[1:])
    if 'pair' in what:
        for c in CASES:
            pair_test(c[0], B, *c[1:])
    if 'benchd' in what:
        for c in CASES:
            bench_bwd(c[0], B, *c[1:], what='d')
    if 'benchw' in what:
        for c in CASES:
            bench_bwd(c[0], B, *c[1:], what='w')


if __name__ == '__main__':
    main()
