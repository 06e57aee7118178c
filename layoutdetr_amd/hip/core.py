"""Thin host-side plumbing between torch tensors (device memory + streams only) and the C ABI."""
import ctypes
import os

import torch

from .. import _lib
from .._lib import Epilogue, Tensor4

ACT_NONE, ACT_RELU, ACT_LRELU = 0, 1, 2

_seed_counter = [0x5EED]

# torch_utils.ops.conv2d_gradfix.no_weight_gradients(): weight gradients of the conv Functions are skipped while this is set
WEIGHT_GRADIENTS_DISABLED = [False]


class _EngineProfile(object):
    """Optional per-launch accounting of the f32-MFMA contraction engine (bench.py roofline leg): HIP events are
    recorded on the launch stream around every engine launch together with its algorithmic FLOP count."""

    def __init__(self):
        self.enabled = False
        self.records = []

    def reset(self):
        self.records = []

    def summary(self):
        """(total algorithmic flops, total seconds, launches) — call after torch.cuda.synchronize()."""
        fl = sum(r[1] for r in self.records)
        ms = sum(r[2].elapsed_time(r[3]) for r in self.records)
        return fl, ms * 1e-3, len(self.records)


PROF = _EngineProfile()


_LEGACY_ENV = ('LDETR_PAIR_D', 'LDETR_DUAL_TRUNK', 'LDETR_FUSED_LAYOUT_LOSSES', 'LDETR_FFN_FUSED', 'LDETR_FFN_FUSED_LARGE', 'LDETR_GROUP_KV', 'LDETR_NO_WORKSPACE',
               'LDETR_TOKEN_STACKS', 'LDETR_SPLIT_BF16', 'LDETR_GEMM_PAIR', 'LDETR_P3_PAIR')
_legacy_warned = [False]


def _warn_legacy_env():
    """The development switches of rounds 1-4 were individual LDETR_* variables; since round 5 they are keys of LDETR_DEBUG="KEY=value,...".  A script that
    still sets an old name would silently run the default path: say so once."""
    if _legacy_warned[0]:
        return
    _legacy_warned[0] = True
    old = [k for k in _LEGACY_ENV if k in os.environ]
    if old:
        import warnings
        warnings.warn(f'{", ".join(old)}: these switches are keys of LDETR_DEBUG now (e.g. LDETR_DEBUG="{old[0][6:]}=0") and are IGNORED as variables; see DESIGN.md, "Diagnostic switches"')


def knob(key, default):
    """Development knob `key` of LDETR_DEBUG="KEY=value,KEY=value" (the one variable the kernel library reads as well: csrc/ldetr_core.cpp), as an int;
    unset -> default (the measured best).  Keys: DESIGN.md, "Diagnostic switches"."""
    _warn_legacy_env()
    for kv in os.environ.get('LDETR_DEBUG', '').split(','):
        k, _, v = kv.partition('=')
        if k.strip() == key and v.strip():
            try:
                return int(v)
            except ValueError:
                return int(float(v))      # the C side parses with atol / atof: accept what it accepts
    return default


def engine_call(tag, flops, thunk, operands=(), nbytes=None):
    """operands: the tensors the launch must read or write at least once (its algorithmic HBM bytes = 4 x their element counts; or `nbytes`)."""
    if not PROF.enabled:
        return thunk()
    s = torch.cuda.Event(enable_timing=True)
    e = torch.cuda.Event(enable_timing=True)
    c0 = engine_launch_counts()
    s.record()
    r = thunk()
    e.record()
    c1 = engine_launch_counts()
    if nbytes is None:
        nbytes = 4.0 * sum(t.numel() for t in operands if t is not None)
    PROF.records.append((tag, float(flops), s, e, nbytes, (c1[0] - c0[0], c1[1] - c0[1]), launched_kernel(tag)))
    return r


_P3_KINDS = {1: 'p3_nt_kernel', 2: 'p3_c3_kernel', 3: 'p3_tn_kernel', 4: 'p3_bwd_pair_nt_kernel', 5: 'p3_bwd_pair_c3_kernel'}
_ENGINE_KINDS = {1: 'gemm_f32_kernel', 2: 'gemm_small_kernel', 3: 'gemm_small_pair_kernel', 4: 'conv3x3_c32_kernel', 5: 'wgrad_c32_3x3_kernel', 6: 'stem_conv7x7_kernel'}


def launched_kernel(tag):
    """Kernel symbol (with the template arguments the launch policy chose) behind the calling thread's most recent engine call of entry point
    `tag`: ldetr_p3_last_launch / ldetr_engine_last_launch.  bench.py labels its per-launch records with it; the parity tests assert it."""
    info = (ctypes.c_int32 * 10)()
    if tag.startswith('ldetr_p3'):
        check(lib().ldetr_p3_last_launch(info), 'p3_last_launch')
        kind, bm, bn, nw, sk = info[0], info[1], info[2], info[3], info[4]
        name = _P3_KINDS.get(kind, 'p3_?')
        if kind in (2, 5):
            return name
        return f'{name}<{bm},{bn},{nw}w>' + (f' splitK={sk}' if sk > 1 else '')
    if tag == 'ldetr_token_stack':
        return 'token-stack kernels (mha_small / mha_cross / ffn / wgrad_multi)'
    check(lib().ldetr_engine_last_launch(info), 'engine_last_launch')
    kind, bm, bn, bk, nw, fast, split, sk = [info[i] for i in range(8)]
    name = _ENGINE_KINDS.get(kind, 'engine_?')
    if kind == 1:
        return f'{name}<{bm},{bn},{bk},{nw}w' + (',FAST' if fast else '') + (',SPLIT' if split else '') + '>' + (f' splitK={sk}' if sk > 1 else '')
    if kind in (2, 3):
        return f'{name}<{nw}w' + (',FAST' if fast else '') + '>'
    if kind == 4:
        return 'conv3x3_c32_split_kernel' if split else name
    return name


def engine_launch_counts():
    """(launches on the f32 MFMA pipe, launches on the bf16 pipe with the exact operand split) issued by this thread so far."""
    a, b = ctypes.c_int64(0), ctypes.c_int64(0)
    lib().ldetr_engine_launch_counts(ctypes.byref(a), ctypes.byref(b))
    return a.value, b.value



_workspace = {}
WORKSPACE_BYTES = int(os.environ.get('LDETR_WORKSPACE_MIB', '256')) << 20   # split-K scratch per device


def disable_splitk_workspace():
    """Unregister the in-kernel split-K scratch on the current device: the engine then reduces split-K with fp32 atomics + an
    epilogue pass (non-deterministic summation order, otherwise equivalent)."""
    l = _lib.load()
    dev = torch.cuda.current_device()
    _workspace[dev] = None
    check(l.ldetr_set_workspace(None, 0), 'set_workspace')


def enable_splitk_workspace():
    """(Re-)register the in-kernel split-K scratch on the current device (the default state)."""
    _workspace.pop(torch.cuda.current_device(), None)
    lib()


def lib():
    l = _lib.load()
    if torch.cuda.is_available():
        dev = torch.cuda.current_device()
        if dev not in _workspace:
            # split-K scratch (arrival counters + partial tiles), registered once per device and kept alive here
            ws = torch.zeros(WORKSPACE_BYTES // 4, dtype=torch.float32, device=torch.device('cuda', dev))
            _workspace[dev] = ws
            check_rc = l.ldetr_set_workspace(ctypes.c_void_p(ws.data_ptr()), ws.numel() * 4)
            if check_rc != 0:
                raise RuntimeError('ldetr_set_workspace failed: ' + l.ldetr_last_error().decode())
    return l


def check(rc, what=''):
    _lib.check(rc, what)


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and t.device.type != 'cuda':
            raise RuntimeError('layoutdetr_amd ops require tensors in GPU memory (no CPU fallback); got device '
                               f'{t.device}')


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr())


def f32c(t):
    """fp32 + contiguous (no copy when already so)."""
    if t is None:
        return None
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def next_seed():
    """Per-launch dropout seed derived from torch's CPU generator (so torch.manual_seed controls it)."""
    _seed_counter[0] += 1
    base = int(torch.initial_seed()) & 0xFFFFFFFF
    return ((base << 32) ^ (_seed_counter[0] * 0x9E3779B1)) & 0xFFFFFFFFFFFFFFFF


_iter_seed = {}


def iter_seed(device=None):
    """Device-resident 64-bit word that every dropout kernel adds to its (host, per-launch) seed.  `reseed()` redraws
    it with torch's graph-safe Philox generator, so a hipGraph replay of the step still sees fresh dropout masks."""
    device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    key = (device.type, device.index)
    if key not in _iter_seed:
        _iter_seed[key] = torch.zeros(1, dtype=torch.int64, device=device)
    return _iter_seed[key]


def reseed(device=None):
    iter_seed(device).random_()


def seed_ptr():
    return ctypes.c_void_p(iter_seed().data_ptr())


def tensor4_nhwc(x):
    """x: [N, H, W, C] tensor (any strides) -> ldetr_tensor4 describing it."""
    N, H, W, C = x.shape
    sn, sh, sw, sc = x.stride()
    return Tensor4(N, C, H, W, sn, sc, sh, sw)


def tensor4_nchw(x):
    """x: [N, C, H, W] tensor (any strides)."""
    N, C, H, W = x.shape
    sn, sc, sh, sw = x.stride()
    return Tensor4(N, C, H, W, sn, sc, sh, sw)


def epilogue(alpha=1.0, col_scale=None, col_bias=None, samp_scale=None, residual=None, act=ACT_NONE, act_alpha=0.0,
             act_gain=1.0, mask_src=None, mask_mode=0, out_scale=1.0, p_drop=0.0, seed=0, accumulate=False, a_rowsum=None):
    ep = Epilogue()
    ep.alpha = alpha
    ep.col_scale = col_scale.data_ptr() if col_scale is not None else None
    ep.col_bias = col_bias.data_ptr() if col_bias is not None else None
    if samp_scale is not None:
        ep.samp_scale = samp_scale.data_ptr()
        ep.samp_ld = samp_scale.stride(0)
    if residual is not None:
        ep.residual = residual.data_ptr()
        ep.ldr = residual.shape[-1]
    ep.act = act
    ep.act_alpha = act_alpha
    ep.act_gain = act_gain
    if mask_src is not None:
        ep.mask_src = mask_src.data_ptr()
        ep.ldm = mask_src.shape[-1]
    ep.mask_mode = mask_mode
    ep.out_scale = out_scale
    ep.p_drop = p_drop
    ep.seed = seed
    ep.seed_ptr = iter_seed().data_ptr() if p_drop > 0 else None
    ep.accumulate = 1 if accumulate else 0
    ep.a_rowsum = a_rowsum.data_ptr() if a_rowsum is not None else None
    return ep


def gemm(A, B, ta, tb, M, N, K, out=None, ep=None, splitk=0, pix_per_sample=0, lda=None, ldb=None):
    """C[M,N] = op(A) @ op(B); see ldetr_gemm_f32.  A/B are 2-D fp32 device tensors (row-major, unit inner stride).
    splitk: 0 = let the launch policy split the reduction when the tile grid would leave CUs idle, 1 = never, >1 = explicit.
    (The library zeroes the output itself when it splits.)"""
    require_gpu(A, B)
    if out is None:
        out = torch.empty((M, N), device=A.device, dtype=torch.float32)
    lda = A.stride(0) if lda is None else lda
    ldb = B.stride(0) if ldb is None else ldb
    engine_call('gemm', 2.0 * M * N * K, lambda: check(lib().ldetr_gemm_f32(
        ptr(A), lda, int(ta), ptr(B), ldb, int(tb), ptr(out), out.stride(0), M, N, K, splitk,
        ctypes.byref(ep) if ep is not None else None, pix_per_sample, stream()), 'gemm'), nbytes=4.0 * (M * K + N * K + M * N))
    return out


def _pair_descs(g0, g1):
    descs = []
    flops = 0.0
    for g in (g0, g1):
        require_gpu(g['A'], g['B'], g['out'])
        d = _lib.GemmDesc()
        d.A = g['A'].data_ptr(); d.lda = g['A'].stride(0); d.ta = int(g['ta'])
        d.B = g['B'].data_ptr(); d.ldb = g['B'].stride(0); d.tb = int(g['tb'])
        d.C = g['out'].data_ptr(); d.ldc = g['out'].stride(0)
        d.M, d.N, d.K, d.splitk = int(g['M']), int(g['N']), int(g['K']), 0
        d.ep = ctypes.addressof(g['ep']) if g.get('ep') is not None else None
        d.pix_per_sample = 0
        descs.append(d)
        flops += 2.0 * d.M * d.N * d.K
    return descs, flops


def gemm_pair_is_single_launch(g0, g1):
    """True if gemm_pair(g0, g1) runs as one kernel launch."""
    descs, _ = _pair_descs(g0, g1)
    return lib().ldetr_gemm_pair_is_single_launch(ctypes.byref(descs[0]), ctypes.byref(descs[1])) == 1


def gemm_pair(g0, g1):
    """Two gemm() calls as one C-ABI call (ldetr_gemm_pair_f32): g = dict(A, B, ta, tb, M, N, K, out, ep).
    The data and the weight gradient of a linear layer (or the two weight gradients of the fused feed-forward block) run as ONE kernel
    launch when both are small-tile problems; otherwise as two, in order."""
    descs, flops = _pair_descs(g0, g1)
    engine_call('gemm', flops, lambda: check(lib().ldetr_gemm_pair_f32(ctypes.byref(descs[0]), ctypes.byref(descs[1]), stream()), 'gemm_pair'),
                nbytes=4.0 * sum(d.M * d.K + d.N * d.K + d.M * d.N for d in descs))


def colsum(a2d, B=1):
    """a2d: [B*P, C] -> [B, C] column sums per group of P rows."""
    require_gpu(a2d)
    R, C = a2d.shape
    if C % 4 != 0:
        # tiny ragged widths (e.g. 3 RGB channels): host reduction of a [R, C] view
        return a2d.reshape(B, R // B, C).sum(1)
    red = torch.zeros((B, C), device=a2d.device, dtype=torch.float32)      # (may be returned as a parameter gradient: never arena memory)
    check(lib().ldetr_colsum_f32(ptr(a2d), ptr(red), B, R // B, C, stream()), 'colsum')
    return red


def flat_grad(p):
    """`p.grad` when p is a Parameter re-homed by training_loop.FlatModule (its .grad is a persistent view of the flat
    gradient buffer): weight-gradient kernels then accumulate straight into it (fp32 atomics / += epilogue) and the
    autograd Function returns None for that input — no temporary, no memset, no separate `grad += dw` launch."""
    if isinstance(p, torch.nn.Parameter) and getattr(p, '_ldetr_flat', False) and p.grad is not None and not torch.is_grad_enabled():
        return p.grad
    return None


class _ZeroArena(object):
    """Small zero-filled accumulation targets (bias / style / demodulation row sums, attention gradient buffers: ~70 `torch.zeros` per iteration,
    one fill launch each) carved from ONE buffer that a single launch re-zeroes at the start of every loss phase (zero_arena_begin, called by
    StyleGAN2Loss.accumulate_gradients).  Only for tensors that die inside the phase that took them and are never handed to autograd as a
    PARAMETER gradient (AccumulateGrad may adopt such a tensor as .grad without copying); without an open phase, during a stream
    capture that found the arena unallocated, or when the arena is full, `zeros` is plain torch.zeros."""

    def __init__(self):
        self.buf = {}          # device index -> (tensor, capacity in floats)
        self.cursor = 0
        self.high = {}         # device index -> floats handed out in the largest phase so far (what begin() re-zeroes, + slack)
        self.open = None       # device index of the open phase

    def begin(self, device):
        dev = device.index if device.index is not None else torch.cuda.current_device()
        self.open = None
        if dev not in self.buf:
            if torch.cuda.is_current_stream_capturing():
                return                      # (allocated outside captures only: a capture's pool must not own a buffer that outlives its graph)
            cap = 1 << 22                   # 16 MiB
            self.buf[dev] = (torch.zeros(cap, device=torch.device('cuda', dev), dtype=torch.float32), cap)
            self.high[dev] = 0
        buf, cap = self.buf[dev]
        n = min(cap, max(self.high[dev] * 2, 1 << 16))
        buf[:n].zero_()
        self.limit, self.cursor, self.open = n, 0, dev

    def take(self, shape, device):
        dev = device.index if device.index is not None else torch.cuda.current_device()
        if self.open != dev:
            return None
        n = 1
        for d in shape:
            n *= int(d)
        pad = (n + 3) // 4 * 4
        if self.cursor + pad > self.limit:
            self.high[dev] = max(self.high[dev], self.cursor + pad)      # the next phase zeroes more
            return None
        out = self.buf[dev][0][self.cursor:self.cursor + n].view(shape)
        self.cursor += pad
        self.high[dev] = max(self.high[dev], self.cursor)
        return out

    def end(self):
        # LDETR_DEBUG="ARENA_GUARD=1": poison what the phase handed out.  An arena view that outlives its phase -- adopted as a leaf .grad, kept on a
        # graph that spans phases -- then shows up as NaN in the next consumer instead of as a silently re-zeroed gradient; the GPU
        # suite is run once with the guard on (profiles/r06_arena_guard.txt).
        if self.open is not None and self.cursor and knob('ARENA_GUARD', 0) and not torch.cuda.is_current_stream_capturing():
            self.buf[self.open][0][:self.cursor].fill_(float('nan'))
        self.open = None


ZERO_ARENA = _ZeroArena()
_ZERO_ARENA_ON = True      # (module switch of the A/B: 0.25 ms per iteration of fill launches at B=16)


def zero_arena_begin(device):
    if _ZERO_ARENA_ON:
        ZERO_ARENA.begin(torch.device(device))


def zero_arena_end():
    ZERO_ARENA.end()


def zeros(shape, device):
    """fp32 zeros for an accumulation target that dies inside the current loss phase (see _ZeroArena)."""
    device = torch.device(device)
    t = ZERO_ARENA.take(tuple(shape), device) if device.type == 'cuda' else None
    return t if t is not None else torch.zeros(shape, device=device, dtype=torch.float32)


def pick_splitk(tiles, K, target=512, min_k=256):
    """Choose a split-K factor so a reduction-heavy GEMM fills the 256 CUs."""
    s = 1
    while tiles * s < target and K // (s * 2) >= min_k and s < 256:
        s *= 2
    return s
