"""The per-iteration G/D step of the reference training loop, MI355X-first.

Reference (training/training_loop.py:274-328), per phase: zero_grad -> requires_grad_(True) (text encoder
re-frozen) -> accumulate_gradients over micro-batches -> `flat = torch.cat(all grads)` -> all_reduce ->
/num_gpus -> nan_to_num(0, 1e5, -1e5) -> split back -> Adam.step(); then G_ema lerp.

Here every module's parameters (and their gradients) live in ONE contiguous fp32 buffer each
(`FlatModule`), so that:
  * zero_grad is one memset, the gradient exchange is an in-place RCCL all-reduce of the flat gradient
    buffer in large buckets on a side HIP stream (no cat / split copies: the reference moves 2x the
    gradient volume through HBM just to pack and unpack),
  * `/world`, nan_to_num and Adam are ONE streaming kernel over (p, g, m, v)  (ldetr_adam_step_f32),
  * the EMA is one kernel over (p_ema, p)  (ldetr_ema_lerp_f32).
Semantics kept: SUM all-reduce then divide by world size, nan_to_num constants, Adam(betas, eps) with
bias correction, EMA beta = 0.5 ** (batch / ema_nimg), buffers of G copied to G_ema.

Everything outside this step (dataset, snapshots, metrics, pickles: training_loop.py:112-171, 341-469)
is out of scope; `training_iteration` is what `bench.py` times.
"""
import copy

import numpy as np
import torch

from ..hip import core
from .networks_detr import split_list  # noqa: F401  (training_loop.py:32 imports it from networks_layoutganpp)


def _phys_view(flat_slice, like):
    """View of a flat slice with the same shape *and memory layout* as the dense tensor `like`."""
    if like.is_contiguous():
        return flat_slice.view(like.shape)
    if like.ndim == 4 and like.is_contiguous(memory_format=torch.channels_last):
        O, I, KH, KW = like.shape
        return flat_slice.view(O, KH, KW, I).permute(0, 3, 1, 2)
    raise RuntimeError('FlatModule: parameter is neither contiguous nor channels_last')


def _trunk_bodies(module):
    from .detr_backbone import ResNet50Body
    return [m for m in module.modules() if isinstance(m, ResNet50Body)]


def refresh_weight_planes(module):
    """The trunk's plane-format weight images (hip/p3.py) follow the parameters: one launch per module after every optimiser step instead of one
    per forward.  Only planes that exist (a body that took the plane-format path at least once) are touched: they are always marked stale --
    whatever LDETR_TRUNK_P3 says now -- and rewritten right away while the plane-format trunk is on (the launch then sits behind the optimiser
    step, outside the captured phases), else lazily by the next forward that needs them."""
    import os
    from .detr_backbone import existing_p3_planes
    on = os.environ.get('LDETR_TRUNK_P3', '1') != '0'
    for body in _trunk_bodies(module):
        planes = existing_p3_planes(body)
        if planes is None:
            continue
        planes.stale = True
        if on:
            planes.managed = True
            planes.ensure()


def mark_weight_planes_stale(module):
    """After any change of the parameters that did not go through DataParallelStep.apply (load / copy / broadcast): torch-level writes are
    also caught by the planes' version check; this covers raw-pointer writes."""
    from .detr_backbone import existing_p3_planes
    for body in _trunk_bodies(module):
        planes = existing_p3_planes(body)
        if planes is not None:
            planes.stale = True


class FlatModule(object):
    """Re-homes all parameters of `module` into one flat fp32 buffer and their .grad into another."""

    def __init__(self, module):
        self.module = module
        # the frozen text encoder (training_loop.py:283 re-freezes it every phase; text_mode 'encoder': ~110 M weights) stays outside the flat
        # buffers: no gradient, no Adam moments, no all-reduce and no optimizer traffic for weights that never change (the reference only
        # flattens parameters whose .grad is not None, :303-305)
        named = [(n, p) for n, p in module.named_parameters() if not n.startswith('text_encoder.')]
        params = [p for _, p in named]
        self.names = [n for n, _ in named]
        self.params = params
        device = params[0].device
        # keep every segment 16-byte aligned so the float4 kernels and conv loaders stay on the vector path
        offs, total = [], 0
        for p in params:
            offs.append(total)
            total += (p.numel() + 3) // 4 * 4
        self.offsets, self.total = offs, total
        self.flat = torch.zeros(total, device=device, dtype=torch.float32)
        self.gflat = torch.zeros(total, device=device, dtype=torch.float32)
        with torch.no_grad():
            for p, off in zip(params, offs):
                view = _phys_view(self.flat[off:off + p.numel()], p.data)
                view.copy_(p.data)
                p.data = view
                p.grad = _phys_view(self.gflat[off:off + p.numel()], p.data)
                p._ldetr_flat = True   # lets the weight-gradient kernels accumulate in place (hip.core.flat_grad)

    def zero_grad(self):
        self.gflat.zero_()
        for p, off in zip(self.params, self.offsets):  # re-attach in case something set .grad = None
            if p.grad is None or p.grad.data_ptr() != self.gflat.data_ptr() + 4 * off:
                p.grad = _phys_view(self.gflat[off:off + p.numel()], p.data)

    def numel(self):
        return self.total

    def stage_segments(self, n_stages=3):
        """Per backward stage (detr_backbone.BackwardStages) the list of (lo, hi) ranges of the flat buffer it completes: stage 1 =
        everything but the trunk (the module's own direct parameters, e.g. D.pos_token, precede `backbone` in parameter order: two
        ranges), stage 2 = trunk layer3 + layer4, stage 3 = stem + layer1 + layer2 — or None when the trunk is not one contiguous
        run of parameters ending in layer3/layer4 (then the phase is exchanged in one piece).  n_stages = 2: stage 2 = the whole trunk."""
        pre = 'backbone.0.body.'
        trunk = [i for i, n in enumerate(self.names) if n.startswith(pre)]
        if not trunk or trunk != list(range(trunk[0], trunk[0] + len(trunk))):
            return None
        late = [i for i in trunk if self.names[i].startswith((pre + 'layer3.', pre + 'layer4.'))]
        if not late or late != list(range(late[0], trunk[-1] + 1)):
            return None
        off = lambda i: self.offsets[i] if i < len(self.offsets) else self.total
        o0, o2, oT = off(trunk[0]), off(late[0]), off(trunk[-1] + 1)
        rest = [r for r in ((0, o0), (oT, self.total)) if r[1] > r[0]]
        if n_stages == 2:
            return [rest, [(o0, oT)]]
        return [rest, [(o2, oT)], [(o0, o2)]]


class Phase(object):
    """One optimiser phase: module + flat Adam state.  'Gmain' / 'Dmain' (or 'Gboth' / 'Dboth' without lazy regularisation) own the state;
    a regulariser phase ('Greg' / 'Dreg', training_loop.py:190-197) is built with `share=<its main phase>`: same flat buffers, same Adam
    moments -- the reference hands ONE optimiser object to both -- and runs every `interval` iterations with gain = interval."""

    def __init__(self, name, module, lr=None, betas=(0.0, 0.99), eps=1e-8, reg_interval=None, share=None, interval=1):
        self.name = name
        self.module = module
        self.interval = int(interval)
        self.main = share if share is not None else self
        if share is not None:
            self.fm, self.m, self.v = share.fm, share.m, share.v
            self.lr, self.betas, self.eps = share.lr, share.betas, share.eps
            return
        self.fm = FlatModule(module)
        if reg_interval is not None:  # lazy-regularisation rescaling, training_loop.py:191-194
            mb_ratio = reg_interval / (reg_interval + 1)
            lr = lr * mb_ratio
            betas = [beta ** mb_ratio for beta in betas]
        self.lr, self.betas, self.eps = lr, tuple(betas), eps
        self.m = torch.zeros_like(self.fm.flat)
        self.v = torch.zeros_like(self.fm.flat)
        self.step = 0
        # torch.optim.Adam skips parameters whose .grad is None and keeps a step count PER PARAMETER.  A regulariser phase reaches only part
        # of the module (R1: what D's conditional score depends on), so after its first step two groups of parameters exist with different
        # step counts: `reg_runs` = the flat ranges the regulariser touches (found once, from its first exchanged gradient), `reg_steps` =
        # how many regulariser steps they have taken on top of `step`.
        self.reg_runs = None
        self.reg_steps = 0

    def touched_runs(self):
        """Flat ranges [(lo, hi)] of the parameters that received a gradient in the regulariser phase that just ran (any non-zero element:
        with the flat gradient buffer 'no gradient' shows as an all-zero segment).  One host synchronisation, once per run of the loop."""
        fm = self.fm
        flags = torch.stack([fm.gflat[o:o + p.numel()].abs().amax() for p, o in zip(fm.params, fm.offsets)]).gt(0).tolist()
        runs = []
        for i, hit in enumerate(flags):
            if not hit:
                continue
            lo = fm.offsets[i]
            hi = fm.offsets[i + 1] if i + 1 < len(fm.offsets) else fm.total
            if runs and runs[-1][1] == lo:
                runs[-1] = (runs[-1][0], hi)
            else:
                runs.append((lo, hi))
        return runs

    def step_ranges(self, regulariser):
        """[(lo, hi, adam step number)] for the optimiser step being taken now (counters already advanced)."""
        main = self.main
        if main.reg_runs is None:
            return [(0, main.fm.total, main.step)]
        if regulariser:
            return [(lo, hi, main.step + main.reg_steps) for lo, hi in main.reg_runs]
        out, pos = [], 0
        for lo, hi in main.reg_runs:
            if lo > pos:
                out.append((pos, lo, main.step))
            out.append((lo, hi, main.step + main.reg_steps))
            pos = hi
        if pos < main.fm.total:
            out.append((pos, main.fm.total, main.step))
        return out


class DataParallelStep(object):
    """Gradient exchange + optimiser for one rank (one process per GPU; RCCL via torch.distributed 'nccl')."""

    def __init__(self, world_size=1, bucket_bytes=256 << 20, fuse_sanitize=True):
        self.world = world_size
        self.bucket = bucket_bytes // 4
        self.fuse = fuse_sanitize
        self.comm_stream = torch.cuda.Stream() if (world_size > 1 and torch.cuda.is_available()) else None
        # diagnostics (bench.py at N > 1): with `record_exposed` set, every apply() brackets its wait for the communication stream with an
        # event pair on the compute stream -> `exposed` holds (phase name, start, end): the time the compute stream stood still for RCCL
        self.record_exposed = False
        self.exposed = []

    def exchange(self, gflat):
        """In-place SUM all-reduce of the flat gradient in large buckets on a side stream (xGMI is point-to-point:
        few large transfers beat many small ones).  Division by world and nan_to_num are fused into Adam."""
        if self.world <= 1:
            return
        import torch.distributed as dist
        n = gflat.numel()
        if self.comm_stream is not None:
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                for s in range(0, n, self.bucket):
                    dist.all_reduce(gflat[s:min(n, s + self.bucket)])
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        else:  # gloo / CPU tests
            for s in range(0, n, self.bucket):
                dist.all_reduce(gflat[s:min(n, s + self.bucket)])

    def collective_plan(self, fm, n_stages=None):
        """The (lo, hi) element ranges of the flat gradient buffer in the order this rank will all-reduce them for one phase: stage by stage
        (FlatModule.stage_segments; None / un-stageable: the whole buffer), each stage's ranges cut into buckets.  A pure function of the
        module's parameter list, the stage count and the bucket size -- every rank must compute the same list (tests/test_host_cpu.py
        checks it at world 2 / 4 / 8 with ranks that MEASURED different stage lengths); a mismatch is a hang in RCCL, not an error."""
        segs = fm.stage_segments(n_stages) if n_stages else None
        ranges = [r for stage in segs for r in stage] if segs is not None else [(0, fm.total)]
        return [(s, min(hi, s + self.bucket)) for lo, hi in ranges for s in range(lo, hi, self.bucket)]

    def agree_min(self, values):
        """Element-wise MIN of a short list of host floats over the ranks (CPU tensor under gloo, device tensor under RCCL).  Every decision that
        shapes the sequence of collectives -- the stage count of the overlapped backward above all -- must come out the same on every rank: a rank
        measuring 1.49 ms where its neighbour measures 1.51 would cut its gradient buffer into different segments."""
        if self.world <= 1 or values is None:
            return values
        import torch.distributed as dist
        dev = 'cuda' if dist.get_backend() == 'nccl' else 'cpu'
        t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return [float(v) for v in t.cpu()]

    def exchange_async(self, gflat, lo, hi):
        """SUM all-reduce of gflat[lo:hi] on the communication stream, ordered after everything queued so far on the current
        stream; the current stream keeps going (the next backward stage).  `finish()` joins."""
        if self.world <= 1 or hi <= lo:
            return
        import torch.distributed as dist
        if self.comm_stream is not None:
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                for s in range(lo, hi, self.bucket):
                    dist.all_reduce(gflat[s:min(hi, s + self.bucket)])
        else:  # gloo / CPU tests: synchronous
            for s in range(lo, hi, self.bucket):
                dist.all_reduce(gflat[s:min(hi, s + self.bucket)])
        self._pending = True

    def finish(self):
        if self.comm_stream is not None and getattr(self, '_pending', False):
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        self._pending = False

    def apply(self, phase, exchanged=False, ema=None):
        """exchanged=True: the caller already reduced the gradient segment by segment (exchange_async); only join here.
        ema = (flat p_ema buffer, beta): the G_ema lerp of training_loop.py:320-328 in the same pass over the parameters (G's parameters do
        not change between this step and the end of the iteration, where the reference updates G_ema)."""
        fm = phase.fm
        ev = None
        if self.record_exposed and self.world > 1 and self.comm_stream is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        if exchanged:
            self.finish()
        else:
            self.exchange(fm.gflat)
        if ev is not None:
            ev[1].record()
            self.exposed.append((phase.name, ev[0], ev[1]))
        main = phase.main
        regulariser = phase is not main
        if regulariser:
            if main.reg_runs is None:
                main.reg_runs = main.touched_runs()     # after the exchange: the same ranges on every rank
            main.reg_steps += 1
        else:
            main.step += 1
        scale = 1.0 / self.world
        if fm.flat.device.type != 'cuda':
            raise RuntimeError('DataParallelStep.apply: parameters must live in GPU memory (no CPU fallback)')
        if not self.fuse:
            core.check(core.lib().ldetr_grad_sanitize_f32(core.ptr(fm.gflat), fm.total, scale, 0.0, 1e5, -1e5, core.stream()), 'grad_sanitize')
        for lo, hi, step in phase.step_ranges(regulariser):
            core.check(core.lib().ldetr_adam_ema_step_f32(core.ptr(fm.flat[lo:hi]), core.ptr(fm.gflat[lo:hi]), core.ptr(main.m[lo:hi]), core.ptr(main.v[lo:hi]), hi - lo,
                                                          step, main.lr, main.betas[0], main.betas[1], main.eps,
                                                          1 if self.fuse else 0, scale, 0.0, 1e5, -1e5,
                                                          core.ptr(ema[0][lo:hi]) if ema is not None else None, float(ema[1]) if ema is not None else 0.0,
                                                          core.stream()), 'adam_step')
        refresh_weight_planes(phase.module)


class EmaTracker(object):
    """G_ema = lerp(G, G_ema, beta) over flat buffers (training_loop.py:320-328)."""

    def __init__(self, G_phase, G_ema):
        self.src = G_phase.fm
        self.G = G_phase.module
        self.G_ema = G_ema
        self.fm = FlatModule(G_ema)
        assert self.fm.total == self.src.total
        for p in G_ema.parameters():
            p.grad = None
            p._ldetr_flat = False
        self.fm.gflat = None
        self._buf_versions = {}

    @staticmethod
    def beta(batch_size, ema_kimg, cur_nimg, ema_rampup=0.05):
        ema_nimg = ema_kimg * 1000
        if ema_rampup is not None:
            ema_nimg = min(ema_nimg, cur_nimg * ema_rampup)
        return 0.5 ** (batch_size / max(ema_nimg, 1e-8))

    def fused(self, phase, batch_size, ema_kimg, cur_nimg, ema_rampup=0.05):
        """-> (p_ema flat buffer, beta) for DataParallelStep.apply(ema=...) when `phase` is the module this tracker follows, else None."""
        if phase.fm is not self.src or batch_size is None or ema_kimg is None:
            return None
        return self.fm.flat, self.beta(batch_size, ema_kimg, cur_nimg, ema_rampup)

    def update(self, batch_size, ema_kimg, cur_nimg, ema_rampup=0.05, lerp_done=False):
        """lerp_done: the parameter lerp already ran inside G's optimiser pass (fused()); only the buffers are synchronised here."""
        if not lerp_done:
            beta = self.beta(batch_size, ema_kimg, cur_nimg, ema_rampup)
            core.check(core.lib().ldetr_ema_lerp_f32(core.ptr(self.fm.flat), core.ptr(self.src.flat), self.src.total, float(beta), core.stream()), 'ema_lerp')
        for i, (b_ema, b) in enumerate(zip(self.G_ema.buffers(), self.G.buffers())):
            ver = (b._version, b.data_ptr())
            if self._buf_versions.get(i) != ver:      # buffers are frozen statistics: copy only when they changed
                b_ema.copy_(b)
                self._buf_versions[i] = ver


def _named_params_and_buffers(module):
    return list(module.named_parameters()) + list(module.named_buffers())     # torch_utils/misc.py:154-156


def _packed_by_dtype(tensors):
    """{dtype: [tensor, ...]} in order of appearance (one collective per dtype instead of one per tensor: a Generator has ~640 tensors)."""
    groups = {}
    for t in tensors:
        groups.setdefault(t.dtype, []).append(t)
    return groups


def broadcast_module(module, src=0):
    """Initial parameter / buffer broadcast (training_loop.py:176-179: one `torch.distributed.broadcast` per tensor there).  Here the
    tensors of one dtype travel as ONE flat buffer: pack -> broadcast -> unpack (xGMI is point-to-point: a 180 MB transfer beats 640
    launches of a few KB).  Same result: every rank ends with rank `src`'s values, layouts untouched."""
    import torch.distributed as dist
    tensors = [t.data for _, t in _named_params_and_buffers(module)]
    for dtype, ts in _packed_by_dtype(tensors).items():
        wire = torch.uint8 if dtype == torch.bool else dtype
        flat = torch.cat([t.reshape(-1).to(wire) for t in ts]) if len(ts) > 1 else ts[0].reshape(-1).to(wire).clone()
        dist.broadcast(flat, src=src)
        off = 0
        for t in ts:
            n = t.numel()
            t.copy_(flat[off:off + n].view(t.shape).to(dtype))
            off += n
    mark_weight_planes_stale(module)


def check_ddp_consistency(module, ignore_regex=None):
    """torch_utils/misc.py:183-194 (called at snapshot time, training_loop.py:402-405): every parameter / buffer must equal rank 0's
    (floats after nan_to_num).  One broadcast per dtype group instead of one per tensor; the failing tensor is named like the
    reference's assert message (`ClassName.parameter.name`)."""
    import re

    import torch.distributed as dist
    named = [(type(module).__name__ + '.' + n, t.detach()) for n, t in _named_params_and_buffers(module)]
    named = [(n, t) for n, t in named if not (ignore_regex is not None and re.fullmatch(ignore_regex, n))]
    by = {}
    for n, t in named:
        by.setdefault(t.dtype, []).append((n, t))
    pairs = []
    for dtype, items in by.items():        # every collective first: a rank that differs must not leave the others waiting in the next one
        wire = torch.uint8 if dtype == torch.bool else dtype
        mine = torch.cat([(torch.nan_to_num(t) if t.is_floating_point() else t).reshape(-1).to(wire) for _, t in items])
        other = mine.clone()
        dist.broadcast(other, src=0)
        pairs.append((items, mine, other))
    for items, mine, other in pairs:
        if not torch.equal(mine, other):
            off = 0
            for n, t in items:
                k = t.numel()
                assert torch.equal(mine[off:off + k], other[off:off + k]), n
                off += k


class StatsCollector(object):
    """The slice of torch_utils/training_stats.py the training loop uses: `report(name, value)` accumulates [count, sum, sum of
    squares] per name on the device (float32 reduction of the reported elements into float64 counters, :91-107) and `update()` sums
    the deltas ACROSS RANKS with one float64 all-reduce of the stacked [names, 3] matrix (:232-254), then exposes mean / std / num of
    the interval since the previous update (`Collector.update`, :158-176).  Names must be reported in the same order on every rank
    (same contract as the reference, :60-64)."""

    def __init__(self, device=None, world=1):
        self.device, self.world = device, world
        self._counters = {}        # name -> float64 [3] on the device
        self._cumulative = {}      # name -> float64 [3] on the host
        self._moments = {}

    def report(self, name, value):
        elems = torch.as_tensor(value)
        if elems.numel() == 0:
            self._counters.setdefault(name, None)
            return value
        # count: known on the host (no launch); sum and sum of squares: one stacked fp32 reduction (the reference's three separate reductions,
        # ones_like and stack were ~8 latency-bound launches per reported name and micro-batch)
        elems = elems.detach().flatten().to(torch.float32)
        m = torch.stack((elems, elems * elems)).sum(1).to(torch.float64)
        c = self._counters.get(name)
        if c is None:
            c = self._counters[name] = torch.zeros(3, dtype=torch.float64, device=elems.device)
        c[0] += float(elems.numel())
        c[1:] += m
        return value

    def update(self):
        names = list(self._counters)
        if not names:
            return {}
        dev = self.device if self.device is not None else torch.device('cpu')
        deltas = torch.stack([(self._counters[n] if self._counters[n] is not None else torch.zeros(3, dtype=torch.float64)).to(dev) for n in names])
        for n in names:
            self._counters[n] = None
        if self.world > 1:
            import torch.distributed as dist
            dist.all_reduce(deltas)
        deltas = deltas.cpu()
        self._moments = {}
        for i, n in enumerate(names):
            self._cumulative[n] = self._cumulative.get(n, torch.zeros(3, dtype=torch.float64)) + deltas[i]
            self._moments[n] = deltas[i]
        return self.as_dict()

    def num(self, name):
        return int(self._moments[name][0]) if name in self._moments else 0

    def mean(self, name):
        d = self._moments.get(name)
        return float(d[1] / d[0]) if d is not None and int(d[0]) != 0 else float('nan')

    def std(self, name):
        d = self._moments.get(name)
        if d is None or int(d[0]) == 0 or not np.isfinite(float(d[1])):
            return float('nan')
        if int(d[0]) == 1:
            return 0.0
        mean = float(d[1] / d[0])
        return float(np.sqrt(max(float(d[2] / d[0]) - mean * mean, 0)))

    def as_dict(self):
        return {n: dict(num=self.num(n), mean=self.mean(n), std=self.std(n)) for n in self._moments}


def copy_params_and_buffers(src_module, dst_module, require_all=False):
    """torch_utils/misc.py:158-165 (resume path, training_loop.py:145-146): by NAME, layouts of the destination kept."""
    src = dict(_named_params_and_buffers(src_module)) if isinstance(src_module, torch.nn.Module) else dict(src_module)
    with torch.no_grad():
        for name, tensor in _named_params_and_buffers(dst_module):
            assert (name in src) or (not require_all), name
            if name in src and tuple(src[name].shape) == tuple(tensor.shape):
                tensor.copy_(src[name].detach().to(tensor.device))
    if isinstance(dst_module, torch.nn.Module):
        mark_weight_planes_stale(dst_module)


def load_pretrained_detr(modules, path='pretrained/up-detr-pre-training-60ep-imagenet.pth', verbose=True):
    """training_loop.py:137-139: `module.load_state_dict(torch.load(path)['model'], strict=False)` for G, D and G_ema — the UP-DETR
    checkpoint initialises the ResNet-50 trunk, `input_proj` and the DETR transformer (same key names here).  The reference
    fails without the file; this path trains from the constructors' initialisation instead and says so (no network to fetch it here)."""
    import os
    path = os.environ.get('LDETR_PRETRAINED', path)
    if not os.path.exists(path):
        if verbose:
            print(f'[layoutdetr_amd] {path} not found: G / D / G_ema keep their random initialisation (the reference loads it with strict=False)')
        return False
    sd = torch.load(path, map_location='cpu')
    sd = sd['model'] if isinstance(sd, dict) and 'model' in sd else sd
    for m in modules:
        own = m.state_dict()
        m.load_state_dict({k: v for k, v in sd.items() if k in own and tuple(own[k].shape) == tuple(v.shape)}, strict=False)
        mark_weight_planes_stale(m)
    return True


def save_snapshot(path, G, D, G_ema, training_set_kwargs, num_gpus=1, rank=0):
    """training_loop.py:395-412: deep copies in eval mode without gradients, cross-rank consistency check (G_ema-style running
    averages excepted) + broadcast from rank 0, moved to the host, pickled by rank 0 as dict(G, D, G_ema, augment_pipe, training_set_kwargs)."""
    import pickle
    data = dict(G=G, D=D, G_ema=G_ema, augment_pipe=None, training_set_kwargs=dict(training_set_kwargs))
    for key in ('G', 'D', 'G_ema'):
        value = _plain_copy(data[key])
        if num_gpus > 1:
            check_ddp_consistency(value, ignore_regex=r'.*\.[^.]+_(avg|ema)')
            broadcast_module(value, src=0)
        data[key] = value.cpu()
    if rank == 0:
        with open(path, 'wb') as f:
            pickle.dump(data, f)
    return data


def _plain_copy(module):
    """copy.deepcopy(module).eval().requires_grad_(False) whose parameters own their memory (the live module's are views into the flat
    step buffers and carry flat .grad views: neither belongs in a snapshot)."""
    grads = [(p, p.grad) for p in module.parameters()]
    for p, _ in grads:
        p.grad = None
    try:
        m = copy.deepcopy(module)
    finally:
        for p, g in grads:
            p.grad = g
    for p in m.parameters():
        p.data = p.data.clone(memory_format=torch.preserve_format)
        if hasattr(p, '_ldetr_flat'):
            p._ldetr_flat = False
    return m.eval().requires_grad_(False)


def load_resume(resume_pkl, G, D, G_ema):
    """training_loop.py:141-146: G / D / G_ema from a snapshot written by save_snapshot() (a pickle of dict(G=, D=, G_ema=) modules) or a
    torch file holding state dicts under the same keys; parameters and buffers are copied by name (require_all=False)."""
    import pickle
    try:
        with open(resume_pkl, 'rb') as f:
            data = pickle.load(f)
    except (pickle.UnpicklingError, ModuleNotFoundError, AttributeError) as e:
        try:
            data = torch.load(resume_pkl, map_location='cpu')
        except Exception:
            raise RuntimeError(f'resume_pkl={resume_pkl!r}: neither a snapshot of this package (save_snapshot) nor a torch file of state dicts; '
                               f'pickles of the reference\'s own classes need the reference tree + its legacy loader ({e})')
    for name, module in (('G', G), ('D', D), ('G_ema', G_ema)):
        if name in data and data[name] is not None:
            copy_params_and_buffers(data[name], module, require_all=False)


def _trunk_body(module):
    bb = getattr(module, 'backbone', None)
    return bb[0].body if bb is not None and hasattr(bb[0], 'body') else None


MIN_STAGE_MS = 1.5      # a backward stage shorter than this cannot hide the host's issue latency between two graph replays (collectives are host-issued)


def backward_stage_count(per_gpu_batch, stage_ms=None):
    """Stages of a phase's backward when the gradient exchange is overlapped with it: layer1-2 | layer3-4 | rest (3) or trunk | rest (2).
    stage_ms: measured durations [rest, layer3-4, layer1-2] of one eager three-stage backward on THIS box (measure_backward_stages): three stages only
    if each of them is long enough (MIN_STAGE_MS) to cover the host issuing the next replay and the collective -- otherwise the exposed communication
    time is host jitter, not bandwidth.  Without a measurement: 2 at <= 4 samples per GPU, else 3.  LDETR_BACKWARD_STAGES = 2 | 3 overrides."""
    import os
    forced = os.environ.get('LDETR_BACKWARD_STAGES')
    if forced in ('2', '3'):
        return int(forced)
    if stage_ms is not None and len(stage_ms) == 3:
        return 3 if min(stage_ms) >= MIN_STAGE_MS else 2
    return 2 if per_gpu_batch <= 4 else 3


def measure_backward_stages(loss, phase, dp, accumulate):
    """One eager forward + three-stage backward of `phase` with HIP events at the stage boundaries -> [ms of stage 1 (incl. the forward), 2, 3], or
    None when the phase has no stageable trunk.  No exchange is issued (timing only); gradients accumulate into the flat buffer as usual."""
    segs = phase.fm.stage_segments(3)
    body = _trunk_body(phase.module)
    if segs is None or body is None or not hasattr(body, 'stages') or not torch.cuda.is_available():
        return None
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    ev[0].record()
    staged_backward(loss, phase, dp, accumulate, between=lambda i: ev[i].record(), exchange=lambda ranges: None, n_stages=3)
    torch.cuda.synchronize()
    return [ev[i].elapsed_time(ev[i + 1]) for i in range(3)]


def staged_backward(loss, phase, dp, run_stage1, between=None, exchange=None, stages=None, n_stages=3):
    """One phase's forward + backward in n_stages (2 or 3) stages with the gradient exchange of each finished segment launched behind it
    (DataParallelStep.exchange_async) -> True if the gradients were exchanged here.  `run_stage1()` runs the forward passes and
    `loss.backward()`; `between(i)` (optional) is called after stage i's work has been queued (graph capture boundaries);
    `exchange(ranges)` replaces the RCCL launch (graph capture: the collectives stay outside the graphs); `stages`: the
    BackwardStages that already recorded this phase's trunk cuts (iteration-level D-trunk sharing evaluates D's trunk before the phases)."""
    from .detr_backbone import BackwardStages
    if stages is not None:
        n_stages = stages.n_stages
    segs = phase.fm.stage_segments(n_stages)
    body = _trunk_body(phase.module)
    if segs is None or body is None or not hasattr(body, 'stages'):
        run_stage1()
        return False
    st = stages if stages is not None else BackwardStages(n_stages)
    if stages is None:
        body.stages = st
    if exchange is None:
        exchange = lambda ranges: [dp.exchange_async(phase.fm.gflat, lo, hi) for lo, hi in ranges]
    try:
        run_stage1()
        if between is not None:
            between(1)
        exchange(segs[0])
        for i in range(2, n_stages + 1):
            st.run(i)
            if between is not None:
                between(i)
            exchange(segs[i - 1])
    finally:
        body.stages = None
    return True


def training_iteration(loss, phases, dp, batch, batch_gpu, gen_z_per_phase, ema=None, batch_size=None, ema_kimg=None, cur_nimg=0, overlap=None,
                       gen_c_per_phase=None, ema_rampup=0.05, batch_idx=0):
    """One iteration = all phases (Gmain, Dmain and -- every `phase.interval`-th iteration, counted by `batch_idx` -- the regulariser phases
    Greg / Dreg with gain = interval) over the rank-local batch, as training_loop.py:274-328.

    batch: dict with bbox_real [b,9,4], bbox_class [b,9], bbox_text (TextFeatures), bbox_patch, padding_mask [b,9] bool,
           background [b,3,R,R], real_c, gen_c.  gen_z_per_phase: list of [b,9,z_dim] tensors, one per phase; gen_c_per_phase: the same for the
           generator's conditioning labels (training_loop.py:257-263 draws one set per phase); None = batch['gen_c'] for every phase.
    """
    b = batch['bbox_real'].shape[0]
    core.reseed(batch['bbox_real'].device)   # fresh device-side dropout seed word for this iteration
    iter_share = getattr(loss, 'share_D_trunk', None) == 'iteration' and any(p.name == 'Dmain' for p in phases)
    if overlap is None:     # overlap the exchange with backward whenever there is an exchange (one micro-batch: the last one is the only one)
        overlap = dp.world > 1
    d_stages = None
    if iter_share:
        # the trunk is cut into backward stages only if the Dmain phase will run them (a recorded cut that is never run() would leave D's
        # trunk without gradient while its flat segment is still reduced and applied)
        d_phase = next(p for p in phases if p.name == 'Dmain')
        if overlap and b <= batch_gpu and d_phase.fm.stage_segments() is not None and hasattr(_trunk_body(d_phase.module), 'stages'):
            from .detr_backbone import BackwardStages
            d_stages = BackwardStages(backward_stage_count(b))
        for s in range(0, b, batch_gpu):
            loss.precompute_D_trunk(batch['background'][s:s + batch_gpu], stages=d_stages)
    lerp_done = False
    due = [batch_idx % getattr(p, 'interval', 1) == 0 for p in phases]
    # the fused G_ema lerp rides on the LAST optimiser step that moves G in this iteration (a Greg phase moves it again after Gmain)
    last_of_fm = {id(p.fm): pi for pi, p in enumerate(phases) if due[pi]}
    for pi, (phase, gen_z) in enumerate(zip(phases, gen_z_per_phase)):
        if not due[pi]:
            continue
        regulariser = getattr(phase, 'main', phase) is not phase
        phase.fm.zero_grad()
        phase.module.requires_grad_(True)
        phase.module.text_encoder.requires_grad_(False)
        gen_c = batch['gen_c'] if gen_c_per_phase is None else gen_c_per_phase[pi]

        def accumulate(phase=phase, gen_z=gen_z, gen_c=gen_c):
            for s in range(0, b, batch_gpu):
                sl = slice(s, s + batch_gpu)
                loss.accumulate_gradients(phase=phase.name, bbox_real=batch['bbox_real'][sl], bbox_class=batch['bbox_class'][sl],
                                          bbox_text=batch['bbox_text'][sl], bbox_patch=batch['bbox_patch'][sl],
                                          padding_mask=batch['padding_mask'][sl], background=batch['background'][sl],
                                          real_c=batch['real_c'][sl], gen_z=gen_z[sl], gen_c=gen_c[sl], gain=getattr(phase, 'interval', 1), cur_nimg=cur_nimg)
        staged = overlap and b <= batch_gpu and not regulariser      # a regulariser phase is exchanged in one piece after its backward
        exchanged = staged_backward(loss, phase, dp, accumulate, stages=(d_stages if (iter_share and phase.name == 'Dmain') else None),
                                    n_stages=backward_stage_count(b)) if staged else (accumulate() or False)
        phase.module.requires_grad_(False)
        fe = ema.fused(phase, batch_size, ema_kimg, cur_nimg, ema_rampup) if (ema is not None and not regulariser and last_of_fm[id(phase.fm)] == pi) else None
        lerp_done = lerp_done or fe is not None
        dp.apply(phase, exchanged=bool(exchanged), ema=fe)
    if ema is not None:
        ema.update(batch_size, ema_kimg, cur_nimg, ema_rampup=ema_rampup, lerp_done=lerp_done)


class GraphedIteration(object):
    """The same iteration with each phase's forward+backward captured once into a hipGraph and replayed.

    A G+D iteration is ~10^4 kernel launches; at 2 samples per GPU (the reference's recommended --gpus=8 --batch=16) the
    eager step is bound by host launch cost, not by the GPU (SURVEY §7 "launch-bound regime").  Requirements, all met by
    the modules here: static shapes (`module.static_shapes = True`: no boolean gathers), no host synchronisation,
    device-side dropout seed word (hip.core.reseed), static input buffers (`self.batch`; copy new data into them).
    The gradient exchange, Adam and EMA stay outside the graphs (RCCL collectives and host-side step counters).
    """

    def __init__(self, loss, phases, dp, batch, batch_gpu, z_dim, ema=None, batch_size=None, ema_kimg=None, capture_stream=None, overlap=None):
        # capture_stream: the side stream the eager warm-up iterations ran on.  Autograd's AccumulateGrad nodes remember the stream of
        # their first use; capturing on that same stream keeps the whole backward on ONE stream (a mismatch makes the engine hop
        # streams inside the capture, and the private-pool allocator then recycles blocks across branches: corrupted replays)
        # overlap (default: world > 1): each phase becomes THREE chained graphs (backward stages, detr_backbone.BackwardStages; two at <= 4 samples
        # per GPU, backward_stage_count); between
        # their replays the finished gradient segment goes to RCCL on the communication stream while the next graph computes.
        self.capture_stream = capture_stream
        self.loss, self.phases, self.dp, self.batch, self.batch_gpu = loss, phases, dp, batch, batch_gpu
        self.ema, self.batch_size, self.ema_kimg = ema, batch_size, ema_kimg
        self.ema_rampup = 0.05
        self.cur_nimg = 0
        self.graphs = []       # per phase: [(graph, flat segment exchanged after it | None)]
        dev = batch['bbox_real'].device
        b = batch['bbox_real'].shape[0]
        self.pre_graph = None
        iter_share = getattr(loss, 'share_D_trunk', None) == 'iteration' and any(p.name == 'Dmain' for p in phases)
        if overlap is None:
            overlap = dp.world > 1
        pool = torch.cuda.graph_pool_handle() if (iter_share or overlap) else None
        # how many stages the overlapped backward gets: from the measured length of the stages on this box, not from a batch-size rule
        self.stage_ms, self.n_stages = None, backward_stage_count(b)
        if overlap and b <= batch_gpu and phases and phases[0].fm.stage_segments() is not None:
            ph = phases[0]
            ph.module.requires_grad_(True); ph.module.text_encoder.requires_grad_(False)

            def once(ph=ph):
                core.reseed(dev)
                gen_z = torch.randn(b, batch['bbox_class'].shape[1], z_dim, device=dev)
                loss.accumulate_gradients(phase=ph.name, bbox_real=batch['bbox_real'], bbox_class=batch['bbox_class'], bbox_text=batch['bbox_text'],
                                          bbox_patch=batch['bbox_patch'], padding_mask=batch['padding_mask'], background=batch['background'],
                                          real_c=batch['real_c'], gen_z=gen_z, gen_c=batch['gen_c'], gain=1, cur_nimg=0)
            cur = torch.cuda.current_stream()
            st = self.capture_stream if self.capture_stream is not None else cur
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                ph.fm.zero_grad()
                self.stage_ms = measure_backward_stages(loss, ph, dp, once)
            cur.wait_stream(st)
            self.stage_ms = dp.agree_min(self.stage_ms)      # one decision for all ranks (every rank takes this branch: same phases, same batch)
            ph.module.requires_grad_(False)
            self.n_stages = backward_stage_count(b, self.stage_ms)
        d_stages = None
        if iter_share:
            # D's trunk forward gets its own graph, replayed before the phases; its activations stay alive in the shared pool until
            # the Dmain graph (captured below, replayed after it) runs the trunk's backward
            d_phase = next(p for p in phases if p.name == 'Dmain')
            if overlap and b <= batch_gpu and d_phase.fm.stage_segments() is not None and hasattr(_trunk_body(d_phase.module), 'stages'):
                from .detr_backbone import BackwardStages
                d_stages = BackwardStages(self.n_stages)
            self.pre_graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.pre_graph, pool=pool, stream=self.capture_stream):
                for s in range(0, b, batch_gpu):
                    loss.precompute_D_trunk(batch['background'][s:s + batch_gpu], stages=d_stages)
        for phase in phases:
            for m in (loss.G, loss.D):
                if not getattr(m, 'static_shapes', False):
                    raise RuntimeError('GraphedIteration needs module.static_shapes = True on G and D')
            phase.fm.zero_grad()
            phase.module.requires_grad_(True)
            phase.module.text_encoder.requires_grad_(False)
            staged = overlap and b <= batch_gpu and phase.fm.stage_segments() is not None
            chain, cur = [], {}

            def begin():
                cur['g'] = torch.cuda.CUDAGraph()
                cur['ctx'] = torch.cuda.graph(cur['g'], pool=pool, stream=self.capture_stream)
                cur['ctx'].__enter__()

            def end(seg):
                cur['ctx'].__exit__(None, None, None)
                chain.append((cur['g'], seg))

            def stage1(phase=phase):
                core.reseed(dev)
                phase.fm.gflat.zero_()
                gen_z = torch.randn(b, batch['bbox_class'].shape[1], z_dim, device=dev)
                for s in range(0, b, batch_gpu):
                    sl = slice(s, s + batch_gpu)
                    loss.accumulate_gradients(phase=phase.name, bbox_real=batch['bbox_real'][sl], bbox_class=batch['bbox_class'][sl],
                                              bbox_text=batch['bbox_text'][sl], bbox_patch=batch['bbox_patch'][sl],
                                              padding_mask=batch['padding_mask'][sl], background=batch['background'][sl],
                                              real_c=batch['real_c'][sl], gen_z=gen_z[sl], gen_c=batch['gen_c'][sl], gain=1, cur_nimg=0)
            begin()
            try:
                if staged:
                    nst = self.n_stages
                    segs = phase.fm.stage_segments(nst)

                    def between(i):
                        end(segs[i - 1])
                        if i < nst:
                            begin()
                    staged_backward(loss, phase, dp, stage1, between=between, exchange=lambda ranges: None,
                                    stages=(d_stages if (iter_share and phase.name == 'Dmain') else None), n_stages=nst)
                else:
                    stage1()
                    end(None)
            except BaseException:
                if cur.get('ctx') is not None and (not chain or chain[-1][0] is not cur['g']):
                    cur['ctx'].__exit__(None, None, None)
                raise
            phase.module.requires_grad_(False)
            self.graphs.append(chain)

    def run(self):
        if self.pre_graph is not None:
            self.pre_graph.replay()
        lerp_done = False
        for phase, chain in zip(self.phases, self.graphs):
            exchanged = False
            for g, seg in chain:
                g.replay()
                if seg is not None:      # this stage's gradient segment is complete: reduce it while the next graph computes
                    for lo, hi in seg:
                        self.dp.exchange_async(phase.fm.gflat, lo, hi)
                    exchanged = True
            fe = self.ema.fused(phase, self.batch_size, self.ema_kimg, self.cur_nimg, self.ema_rampup) if self.ema is not None else None
            lerp_done = lerp_done or fe is not None
            self.dp.apply(phase, exchanged=exchanged, ema=fe)
        if self.ema is not None:
            self.ema.update(self.batch_size, self.ema_kimg, self.cur_nimg, ema_rampup=self.ema_rampup, lerp_done=lerp_done)
        if self.batch_size:
            self.cur_nimg += self.batch_size


# ---------------------------------------------------------------------------------------------------------------------
# training_loop(): the reference's entry point (training/training_loop.py:63-100), same keyword arguments, driving the step above.

def construct_class_by_name(*args, class_name=None, **kwargs):
    """dnnlib.util.construct_class_by_name (dnnlib/util.py:302-304): 'pkg.module.Class' -> Class(*args, **kwargs)."""
    import importlib
    mod, _, name = class_name.rpartition('.')
    return getattr(importlib.import_module(mod), name)(*args, **kwargs)


class InfiniteSampler(torch.utils.data.Sampler):
    """torch_utils/misc.py:118-150: an endless shuffled index stream; rank r takes every index whose position is r mod W."""

    def __init__(self, dataset, rank=0, num_replicas=1, shuffle=True, seed=0, window_size=0.5):
        self.n, self.rank, self.world, self.shuffle, self.seed, self.window = len(dataset), rank, num_replicas, shuffle, seed, window_size

    def __iter__(self):
        order = np.arange(self.n)
        rnd, window = None, 0
        if self.shuffle:
            rnd = np.random.RandomState(self.seed)
            rnd.shuffle(order)
            window = int(np.rint(order.size * self.window))
        idx = 0
        while True:
            i = idx % order.size
            if idx % self.world == self.rank:
                yield order[i]
            if window >= 2:
                j = (i - rnd.randint(window)) % order.size
                order[i], order[j] = order[j], order[i]
            idx += 1


def training_loop(run_dir='.', training_set_kwargs={}, validation_set_kwargs={}, data_loader_kwargs={}, G_kwargs={}, D_kwargs={},
                  G_opt_kwargs={}, D_opt_kwargs={}, augment_kwargs=None, loss_kwargs={}, metrics=[], random_seed=0, num_gpus=1, rank=0,
                  batch_size=4, batch_gpu=4, ema_kimg=10, ema_rampup=0.05, G_reg_interval=None, D_reg_interval=16, augment_p=0,
                  ada_target=None, ada_interval=4, ada_kimg=500, total_kimg=25000, kimg_per_tick=4, image_snapshot_ticks=50,
                  network_snapshot_ticks=50, resume_pkl=None, resume_kimg=0, cudnn_benchmark=True, abort_fn=None, progress_fn=None):
    """Same keyword arguments as the reference's `training_loop` (train.py:47 calls it with `**c`), so `train.py` drives this one
    unchanged (through layoutdetr_amd.dropin.install() or by importing it).  What it runs: dataset + DataLoader + InfiniteSampler
    exactly as :112-118, networks / loss by class name (:127-135,158), then per iteration the flat-parameter step of this module
    (phases Gmain / Dmain; the reg phases are no-ops at r1_gamma = pl_weight = 0, which is all `train.py` configures: :135-136).
    Also as the reference: the UP-DETR checkpoint into G / D / G_ema when `pretrained/up-detr-pre-training-60ep-imagenet.pth` exists
    (:137-139), `resume_pkl` (:140-146), one flat broadcast per module (:176-179), per-phase latents AND conditioning labels (:257-263),
    `ema_rampup` (:321-323), per-tick statistics summed over ranks in float64 + `stats.jsonl` (:428-447), network snapshots with the
    cross-rank consistency check (:395-412).  NOT done here (a note is printed): augment pipe, ADA, image snapshots, metric evaluation,
    tensorboard — SURVEY §8 marks them outside the hot path.  Returns dict(stats of the last tick, G, D, G_ema, snapshot_pkl)."""
    import json
    import os
    import time
    assert augment_kwargs is None and ada_target is None, 'the augment pipe / ADA are not part of the hot path'
    start_time = time.time()
    device = torch.device('cuda', rank)
    torch.cuda.set_device(device)
    np.random.seed(random_seed * num_gpus + rank)
    torch.manual_seed(random_seed * num_gpus + rank)
    if rank == 0:
        print('Loading training set...')
    training_set = construct_class_by_name(**training_set_kwargs)
    sampler = InfiniteSampler(dataset=training_set, rank=rank, num_replicas=num_gpus, seed=random_seed)
    dl_kwargs = dict(data_loader_kwargs)
    if getattr(training_set, 'mode', None) == 'device' and hasattr(training_set, 'collate'):
        dl_kwargs.setdefault('collate_fn', training_set.collate)     # uint8 pages + 0-stride patch placeholder (dataset_layoutganpp.LayoutDataset)
    it = iter(torch.utils.data.DataLoader(dataset=training_set, sampler=sampler, batch_size=batch_size // num_gpus, **dl_kwargs))
    common = dict(num_bbox_labels=training_set.num_bbox_labels, img_channels=training_set.num_channels, img_height=training_set.height,
                  img_width=training_set.width, background_size=training_set.background_size_for_training, c_dim=training_set.label_dim)
    if rank == 0:
        print('Constructing networks...')
    G = construct_class_by_name(**G_kwargs, **common).train().requires_grad_(False).to(device)
    D = construct_class_by_name(**D_kwargs, **common).train().requires_grad_(False).to(device)
    G_ema = copy.deepcopy(G).eval()
    load_pretrained_detr((G, D, G_ema), verbose=(rank == 0))          # :137-139
    if resume_pkl is not None and rank == 0:                          # :140-146 (rank 0 loads; the broadcast below distributes)
        print(f'Resuming from "{resume_pkl}"')
        load_resume(resume_pkl, G, D, G_ema)
    if num_gpus > 1:
        if rank == 0:
            print(f'Distributing across {num_gpus} GPUs...')
        for module in (G, D, G_ema):
            broadcast_module(module, src=0)          # :176-179
    collector = StatsCollector(device=device, world=num_gpus)         # training_stats.init_multiprocessing + Collector (:111, :204-206)
    lk = dict(loss_kwargs)
    loss = construct_class_by_name(device=device, G=G, D=D, augment_pipe=None, report_fn=collector.report, **lk)
    phases = []
    for name, module, opt, reg in (('G', G, G_opt_kwargs, G_reg_interval), ('D', D, D_opt_kwargs, D_reg_interval)):
        opt = dict(opt)
        opt.pop('class_name', None)                   # torch.optim.Adam in the reference; the fused Adam kernel here
        phases.append(Phase(name + ('both' if reg is None else 'main'), module, lr=opt.get('lr', 1e-3), betas=tuple(opt.get('betas', (0.9, 0.999))),
                            eps=opt.get('eps', 1e-8), reg_interval=reg))
        # :195-197: the lazy regulariser phase shares the main phase's optimiser.  Built only when its regulariser is on (pl_weight for G,
        # r1_gamma for D): with the weight at 0 the reference's phase is a no-op that leaves every .grad None, so its Adam step changes nothing
        if reg is not None and float(getattr(loss, 'pl_weight' if name == 'G' else 'r1_gamma', 0) or 0) != 0:
            phases.append(Phase(name + 'reg', module, share=phases[-1], interval=reg))
    dp = DataParallelStep(world_size=num_gpus)
    ema = EmaTracker(phases[0], G_ema)
    if rank == 0:
        print('Not run on this path: augment pipe, ADA, image snapshots, metrics (outside the hot path)')
        print(f'Training for {total_kimg} kimg...')
    cur_nimg, cur_tick, tick_start_nimg, tick_start = resume_kimg * 1000, 0, resume_kimg * 1000, time.time()
    batch_idx = 0
    if progress_fn is not None:
        progress_fn(0, total_kimg)
    last = {}
    stats_jsonl = open(os.path.join(run_dir, 'stats.jsonl'), 'wt') if (rank == 0 and run_dir and os.path.isdir(run_dir)) else None     # :200-202
    snapshot_pkl = None
    labelled = training_set.label_dim > 0 and hasattr(training_set, 'get_label')
    while True:
        samples, real_c = next(it)
        texts = list(map(list, zip(*samples['texts']))) if isinstance(samples.get('texts'), (list, tuple)) else samples['texts']   # :246
        b = samples['bboxes'].shape[0]
        if isinstance(texts, list):     # strings -> tokens ONCE per iteration (the reference re-tokenises inside each of the 5 G/D forwards)
            from .networks_detr import _coerce_text
            texts = _coerce_text(G, texts, device)
        from .dataset_layoutganpp import batch_backgrounds_to_device, patch_placeholder_to_device
        batch = dict(bbox_real=samples['bboxes'].to(device).float(), bbox_class=samples['labels'].to(device).long(), bbox_text=texts,
                     bbox_patch=patch_placeholder_to_device(samples['patches'], device), padding_mask=~samples['mask'].to(device).bool(),
                     background=batch_backgrounds_to_device(samples['background'], training_set.background_size_for_training, device),
                     real_c=real_c.to(device))
        # :257-263: one set of latents AND one set of conditioning labels (labels of random dataset items) PER PHASE
        gen_z = [torch.randn(b, batch['bbox_class'].shape[1], G.z_dim, device=device) for _ in phases]
        if labelled:
            gen_c = [torch.from_numpy(np.stack([training_set.get_label(np.random.randint(len(training_set))) for _ in range(b)])).to(device) for _ in phases]
        else:
            gen_c = [torch.zeros_like(batch['real_c']) for _ in phases]
        batch['gen_c'] = gen_c[0]
        training_iteration(loss, phases, dp, batch, batch_gpu, gen_z, ema=ema, batch_size=batch_size, ema_kimg=ema_kimg, cur_nimg=cur_nimg,
                           gen_c_per_phase=gen_c, ema_rampup=ema_rampup, batch_idx=batch_idx)
        cur_nimg += batch_size
        batch_idx += 1
        done = cur_nimg >= total_kimg * 1000
        if not done and cur_tick != 0 and cur_nimg < tick_start_nimg + kimg_per_tick * 1000:
            continue
        torch.cuda.synchronize()
        now = time.time()
        # :428-447: per-tick statistics, summed over ranks in float64 (one all-reduce), jsonl line by rank 0
        tick_stats = collector.update()
        last = {k: v['mean'] for k, v in tick_stats.items()}
        if rank == 0:
            print(f'tick {cur_tick:<5d} kimg {cur_nimg / 1e3:<8.1f} sec/kimg {(now - tick_start) / max(cur_nimg - tick_start_nimg, 1) * 1e3:<7.2f} ' +
                  ' '.join(f'{k.split("/")[-1]} {v:.3f}' for k, v in sorted(last.items()) if k.startswith('Loss/scores')))
            if stats_jsonl is not None:
                stats_jsonl.write(json.dumps(dict(tick_stats, timestamp=time.time(), **{'Progress/kimg': dict(num=1, mean=cur_nimg / 1e3, std=0.0)})) + '\n')
                stats_jsonl.flush()
        if abort_fn is not None and abort_fn():
            done = True
        # :395-412: network snapshot (every `network_snapshot_ticks` ticks and at the end), consistency-checked across ranks
        if network_snapshot_ticks is not None and (done or cur_tick % network_snapshot_ticks == 0) and run_dir and os.path.isdir(run_dir):
            snapshot_pkl = os.path.join(run_dir, f'network-snapshot-{cur_nimg // 1000:06d}.pkl')
            save_snapshot(snapshot_pkl, G, D, G_ema, training_set_kwargs, num_gpus=num_gpus, rank=rank)
        cur_tick += 1
        tick_start_nimg, tick_start = cur_nimg, time.time()
        if progress_fn is not None:
            progress_fn(cur_nimg // 1000, total_kimg)
        if done:
            break
    if stats_jsonl is not None:
        stats_jsonl.close()
    last['cur_nimg'] = cur_nimg
    last['total_sec'] = time.time() - start_time
    return dict(stats=last, G=G, D=D, G_ema=G_ema, snapshot_pkl=snapshot_pkl, stats_detail=tick_stats)
