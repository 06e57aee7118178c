"""Per-block timeline of one launch of the LDS-tiled contraction kernel (development aid): when blocks start, how long the
prologue / main loop / epilogue (+ split-K fix-up) take, and how long the launch spans.  usage: python tools/trace_tiles.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from layoutdetr_amd.hip import core
dev = torch.device('cuda:0')
L = core.lib()
def conv(N, H, Ci, Co, k, s, p, what='fwd'):
    OH = (H + 2 * p - k) // s + 1
    x = torch.randn(N, H, H, Ci, device=dev); w = torch.randn(Co, k, k, Ci, device=dev); dy = torch.randn(N, OH, OH, Co, device=dev)
    y = torch.empty(N, OH, OH, Co, device=dev); dx = torch.empty_like(x)
    xt = core.tensor4_nhwc(x); dyt = core.tensor4_nhwc(dy)
    sc = torch.rand(Co, device=dev) + 0.5; sh = torch.randn(Co, device=dev); res = torch.randn_like(y)
    ep = core.epilogue(col_scale=sc, col_bias=sh, residual=res.reshape(-1, Co), act=core.ACT_RELU)
    if what == 'fwd':
        return lambda: L.ldetr_conv2d_fwd_f32(core.ptr(x), ctypes.byref(xt), core.ptr(w), Co, k, k, s, p, core.ptr(y), Co, OH, OH, None, 0, ctypes.byref(ep), core.stream())
    return lambda: L.ldetr_conv2d_bwd_data_f32(core.ptr(dy), ctypes.byref(dyt), core.ptr(w), Ci, k, k, s, p, core.ptr(dx), Ci, H, H, None, 0, None, core.stream())
buf = torch.zeros(5 * 65536, dtype=torch.int64, device=dev)
CASES = [('l3 3x3 256->256 fwd', (16, 16, 256, 256, 3, 1, 1, 'fwd')), ('l2 3x3 128->128 fwd', (16, 32, 128, 128, 3, 1, 1, 'fwd')),
                   ('l3 1x1 256->1024 fwd', (16, 16, 256, 1024, 1, 1, 0, 'fwd')), ('l4 1x1 512->2048 bwdD', (16, 8, 512, 2048, 1, 1, 0, 'bwd')),
                   ('sg 3x3 128->128 @64 fwd', (16, 64, 128, 128, 3, 1, 1, 'fwd'))]
if len(sys.argv) > 1 and sys.argv[1] == '1x1':
    CASES = [('l1 1x1 64->256 fwd', (16, 64, 64, 256, 1, 1, 0, 'fwd')), ('l1 1x1 256->64 fwd', (16, 64, 256, 64, 1, 1, 0, 'fwd')), ('l2 1x1 256->128 fwd', (16, 64, 256, 128, 1, 1, 0, 'fwd')),
             ('l2 1x1 128->512 fwd', (16, 32, 128, 512, 1, 1, 0, 'fwd')), ('l3 1x1 1024->256 fwd', (16, 16, 1024, 256, 1, 1, 0, 'fwd')), ('l2 1x1 256->128 bwdD', (16, 64, 256, 128, 1, 1, 0, 'bwd'))]
for name, args in CASES:
    f = conv(*args)
    for _ in range(3): f()
    torch.cuda.synchronize()
    buf.zero_(); torch.cuda.synchronize()
    L.ldetr_debug_trace_tiles(ctypes.c_void_p(buf.data_ptr()))
    f(); torch.cuda.synchronize()
    L.ldetr_debug_trace_tiles(None)
    raw = buf.view(-1, 5).cpu(); raw = raw[raw[:, 3] > 0]
    if len(raw) == 0:
        print(f'{name:26s} not routed to the LDS-tiled kernel'); continue
    place = ((raw[:, 4] >> 32) & 0xF) * 65536 + (raw[:, 4] & 0xFF00)      # XCC, SE / SH / CU bits of HW_ID
    per_cu = torch.unique(place, return_counts=True)[1]
    t = raw[:, :4].double() * 0.01        # us
    t0 = t[:, 0].min()
    start = t[:, 0] - t0; end = t[:, 3] - t0
    q = lambda v: ' '.join(f'{float(v.quantile(x)):6.1f}' for x in (0.0, 0.5, 0.9, 1.0))
    loop = (raw[:, 2] - raw[:, 1]).double() * 0.01
    xcc = ((raw[:, 4] >> 32) & 0xF)
    by_xcc = [round(float(loop[xcc == x].mean()), 1) for x in sorted(set(xcc.tolist()))]
    nb = len(raw); idx = torch.arange(nb)
    quarters = [round(float(loop[(idx * 4 // nb) == qq].mean()), 1) for qq in range(4)]
    cu_mean = {}
    for pl, lv in zip(place.tolist(), loop.tolist()): cu_mean.setdefault(pl, []).append(lv)
    cu_means = torch.tensor([sum(v) / len(v) for v in cu_mean.values()])
    print(f'    loop mean by XCC {by_xcc} | by quarter of the tile order {quarters} | per-CU mean min/median/max {float(cu_means.min()):.1f}/{float(cu_means.median()):.1f}/{float(cu_means.max()):.1f}')
    print(f'{name:26s} blocks {len(t):5d} | start  {q(start)} | prologue {q(t[:, 1] - t[:, 0])} | loop {q(t[:, 2] - t[:, 1])} | tail {q(t[:, 3] - t[:, 2])} | end {q(end)}   (us; min/median/p90/max) | CUs used {len(per_cu)}, blocks per CU {int(per_cu.min())}..{int(per_cu.max())} hist {torch.bincount(per_cu).tolist()}', flush=True)
