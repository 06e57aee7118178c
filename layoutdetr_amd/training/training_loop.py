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

    def stage_segments(self):
        """Per backward stage (detr_backbone.BackwardStages) the list of (lo, hi) ranges of the flat buffer it completes: stage 1 =
        everything but the trunk (the module's own direct parameters, e.g. D.pos_token, precede `backbone` in parameter order: two
        ranges), stage 2 = trunk layer3 + layer4, stage 3 = stem + layer1 + layer2 — or None when the trunk is not one contiguous
        run of parameters ending in layer3/layer4 (then the phase is exchanged in one piece)."""
        pre = 'backbone.0.body.'
        trunk = [i for i, n in enumerate(self.names) if n.startswith(pre)]
        if not trunk or trunk != list(range(trunk[0], trunk[0] + len(trunk))):
            return None
        late = [i for i in trunk if self.names[i].startswith((pre + 'layer3.', pre + 'layer4.'))]
        if not late or late != list(range(late[0], trunk[-1] + 1)):
            return None
        off = lambda i: self.offsets[i] if i < len(self.offsets) else self.total
        o0, o2, oT = off(trunk[0]), off(late[0]), off(trunk[-1] + 1)
        return [[r for r in ((0, o0), (oT, self.total)) if r[1] > r[0]], [(o2, oT)], [(o0, o2)]]


class Phase(object):
    """One optimiser phase ('Gmain' or 'Dmain'): module + flat Adam state."""

    def __init__(self, name, module, lr, betas=(0.0, 0.99), eps=1e-8, reg_interval=None):
        self.name = name
        self.module = module
        self.fm = FlatModule(module)
        if reg_interval is not None:  # lazy-regularisation rescaling, training_loop.py:191-194
            mb_ratio = reg_interval / (reg_interval + 1)
            lr = lr * mb_ratio
            betas = [beta ** mb_ratio for beta in betas]
        self.lr, self.betas, self.eps = lr, tuple(betas), eps
        self.m = torch.zeros_like(self.fm.flat)
        self.v = torch.zeros_like(self.fm.flat)
        self.step = 0


class DataParallelStep(object):
    """Gradient exchange + optimiser for one rank (one process per GPU; RCCL via torch.distributed 'nccl')."""

    def __init__(self, world_size=1, bucket_bytes=256 << 20, fuse_sanitize=True):
        self.world = world_size
        self.bucket = bucket_bytes // 4
        self.fuse = fuse_sanitize
        self.comm_stream = torch.cuda.Stream() if (world_size > 1 and torch.cuda.is_available()) else None

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

    def apply(self, phase, exchanged=False):
        """exchanged=True: the caller already reduced the gradient segment by segment (exchange_async); only join here."""
        fm = phase.fm
        core.join_side()   # weight-gradient launches of a backward that was not run through Loss.accumulate_gradients
        if exchanged:
            self.finish()
        else:
            self.exchange(fm.gflat)
        phase.step += 1
        scale = 1.0 / self.world
        if fm.flat.device.type != 'cuda':
            raise RuntimeError('DataParallelStep.apply: parameters must live in GPU memory (no CPU fallback)')
        if not self.fuse:
            core.check(core.lib().ldetr_grad_sanitize_f32(core.ptr(fm.gflat), fm.total, scale, 0.0, 1e5, -1e5, core.stream()), 'grad_sanitize')
        core.check(core.lib().ldetr_adam_step_f32(core.ptr(fm.flat), core.ptr(fm.gflat), core.ptr(phase.m), core.ptr(phase.v), fm.total,
                                                  phase.step, phase.lr, phase.betas[0], phase.betas[1], phase.eps,
                                                  1 if self.fuse else 0, scale, 0.0, 1e5, -1e5, core.stream()), 'adam_step')


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

    def update(self, batch_size, ema_kimg, cur_nimg, ema_rampup=0.05):
        ema_nimg = ema_kimg * 1000
        if ema_rampup is not None:
            ema_nimg = min(ema_nimg, cur_nimg * ema_rampup)
        beta = 0.5 ** (batch_size / max(ema_nimg, 1e-8))
        core.check(core.lib().ldetr_ema_lerp_f32(core.ptr(self.fm.flat), core.ptr(self.src.flat), self.src.total, float(beta), core.stream()), 'ema_lerp')
        for i, (b_ema, b) in enumerate(zip(self.G_ema.buffers(), self.G.buffers())):
            ver = (b._version, b.data_ptr())
            if self._buf_versions.get(i) != ver:      # buffers are frozen statistics: copy only when they changed
                b_ema.copy_(b)
                self._buf_versions[i] = ver


def broadcast_module(module, src=0):
    """Initial parameter/buffer broadcast (training_loop.py:176-179) — one collective per flat buffer when available."""
    import torch.distributed as dist
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src)


def _trunk_body(module):
    bb = getattr(module, 'backbone', None)
    return bb[0].body if bb is not None and hasattr(bb[0], 'body') else None


def staged_backward(loss, phase, dp, run_stage1, between=None, exchange=None, stages=None):
    """One phase's forward + backward in three stages with the gradient exchange of each finished segment launched behind it
    (DataParallelStep.exchange_async) -> True if the gradients were exchanged here.  `run_stage1()` runs the forward passes and
    `loss.backward()`; `between(i)` (optional) is called after stage i's work has been queued (graph capture boundaries);
    `exchange(ranges)` replaces the RCCL launch (graph capture: the collectives stay outside the graphs); `stages`: the
    BackwardStages that already recorded this phase's trunk cuts (iteration-level D-trunk sharing evaluates D's trunk before the phases)."""
    from .detr_backbone import BackwardStages
    segs = phase.fm.stage_segments()
    body = _trunk_body(phase.module)
    if segs is None or body is None or not hasattr(body, 'stages'):
        run_stage1()
        return False
    st = stages if stages is not None else BackwardStages()
    if stages is None:
        body.stages = st
    if exchange is None:
        exchange = lambda ranges: [dp.exchange_async(phase.fm.gflat, lo, hi) for lo, hi in ranges]
    try:
        run_stage1()
        core.join_side()
        if between is not None:
            between(1)
        exchange(segs[0])
        for i in (2, 3):
            st.run(i)
            core.join_side()
            if between is not None:
                between(i)
            exchange(segs[i - 1])
    finally:
        body.stages = None
    return True


def training_iteration(loss, phases, dp, batch, batch_gpu, gen_z_per_phase, ema=None, batch_size=None, ema_kimg=None, cur_nimg=0, overlap=None):
    """One iteration = all phases (Gmain, Dmain) over the rank-local batch, as training_loop.py:274-328.

    batch: dict with bbox_real [b,9,4], bbox_class [b,9], bbox_text (TextFeatures), bbox_patch, padding_mask [b,9] bool,
           background [b,3,R,R], real_c, gen_c.  gen_z_per_phase: list of [b,9,z_dim] tensors, one per phase.
    """
    b = batch['bbox_real'].shape[0]
    core.reseed(batch['bbox_real'].device)   # fresh device-side dropout seed word for this iteration
    iter_share = getattr(loss, 'share_D_trunk', None) == 'iteration' and any(p.name == 'Dmain' for p in phases)
    if overlap is None:     # overlap the exchange with backward whenever there is an exchange (one micro-batch: the last one is the only one)
        overlap = dp.world > 1
    d_stages = None
    if iter_share:
        if overlap and b <= batch_gpu:
            from .detr_backbone import BackwardStages
            d_stages = BackwardStages()
        for s in range(0, b, batch_gpu):
            loss.precompute_D_trunk(batch['background'][s:s + batch_gpu], stages=d_stages)
    for phase, gen_z in zip(phases, gen_z_per_phase):
        phase.fm.zero_grad()
        phase.module.requires_grad_(True)
        phase.module.text_encoder.requires_grad_(False)

        def accumulate(phase=phase, gen_z=gen_z):
            for s in range(0, b, batch_gpu):
                sl = slice(s, s + batch_gpu)
                loss.accumulate_gradients(phase=phase.name, bbox_real=batch['bbox_real'][sl], bbox_class=batch['bbox_class'][sl],
                                          bbox_text=batch['bbox_text'][sl], bbox_patch=batch['bbox_patch'][sl],
                                          padding_mask=batch['padding_mask'][sl], background=batch['background'][sl],
                                          real_c=batch['real_c'][sl], gen_z=gen_z[sl], gen_c=batch['gen_c'][sl], gain=1, cur_nimg=cur_nimg)
        staged = overlap and b <= batch_gpu
        exchanged = staged_backward(loss, phase, dp, accumulate, stages=(d_stages if (iter_share and phase.name == 'Dmain') else None)) if staged else (accumulate() or False)
        phase.module.requires_grad_(False)
        dp.apply(phase, exchanged=True) if exchanged else dp.apply(phase)
    if ema is not None:
        ema.update(batch_size, ema_kimg, cur_nimg)


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
        # overlap (default: world > 1): each phase becomes THREE chained graphs (backward stages, detr_backbone.BackwardStages); between
        # their replays the finished gradient segment goes to RCCL on the communication stream while the next graph computes.
        self.capture_stream = capture_stream
        self.loss, self.phases, self.dp, self.batch, self.batch_gpu = loss, phases, dp, batch, batch_gpu
        self.ema, self.batch_size, self.ema_kimg = ema, batch_size, ema_kimg
        self.cur_nimg = 0
        self.graphs = []       # per phase: [(graph, flat segment exchanged after it | None)]
        dev = batch['bbox_real'].device
        b = batch['bbox_real'].shape[0]
        self.pre_graph = None
        iter_share = getattr(loss, 'share_D_trunk', None) == 'iteration' and any(p.name == 'Dmain' for p in phases)
        if overlap is None:
            overlap = dp.world > 1
        pool = torch.cuda.graph_pool_handle() if (iter_share or overlap) else None
        d_stages = None
        if iter_share:
            # D's trunk forward gets its own graph, replayed before the phases; its activations stay alive in the shared pool until
            # the Dmain graph (captured below, replayed after it) runs the trunk's backward
            if overlap and b <= batch_gpu:
                from .detr_backbone import BackwardStages
                d_stages = BackwardStages()
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
                    segs = phase.fm.stage_segments()

                    def between(i):
                        end(segs[i - 1])
                        if i < 3:
                            begin()
                    staged_backward(loss, phase, dp, stage1, between=between, exchange=lambda ranges: None,
                                    stages=(d_stages if (iter_share and phase.name == 'Dmain') else None))
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
        for phase, chain in zip(self.phases, self.graphs):
            exchanged = False
            for g, seg in chain:
                g.replay()
                if seg is not None:      # this stage's gradient segment is complete: reduce it while the next graph computes
                    for lo, hi in seg:
                        self.dp.exchange_async(phase.fm.gflat, lo, hi)
                    exchanged = True
            self.dp.apply(phase, exchanged=True) if exchanged else self.dp.apply(phase)
        if self.ema is not None:
            self.ema.update(self.batch_size, self.ema_kimg, self.cur_nimg)
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
    Out of this path's scope and therefore NOT done here (a note is printed): augment pipe, ADA, image / network snapshots, metric
    evaluation, tensorboard — SURVEY §8 marks them outside the hot path.  Returns dict(stats of the last tick, G, D, G_ema)."""
    import time
    assert augment_kwargs is None and ada_target is None, 'the augment pipe / ADA are not part of the hot path'
    device = torch.device('cuda', rank)
    torch.cuda.set_device(device)
    np.random.seed(random_seed * num_gpus + rank)
    torch.manual_seed(random_seed * num_gpus + rank)
    if rank == 0:
        print('Loading training set...')
    training_set = construct_class_by_name(**training_set_kwargs)
    sampler = InfiniteSampler(dataset=training_set, rank=rank, num_replicas=num_gpus, seed=random_seed)
    it = iter(torch.utils.data.DataLoader(dataset=training_set, sampler=sampler, batch_size=batch_size // num_gpus, **data_loader_kwargs))
    common = dict(num_bbox_labels=training_set.num_bbox_labels, img_channels=training_set.num_channels, img_height=training_set.height,
                  img_width=training_set.width, background_size=training_set.background_size_for_training, c_dim=training_set.label_dim)
    if rank == 0:
        print('Constructing networks...')
    G = construct_class_by_name(**G_kwargs, **common).train().requires_grad_(False).to(device)
    D = construct_class_by_name(**D_kwargs, **common).train().requires_grad_(False).to(device)
    if num_gpus > 1:
        for module in (G, D):
            broadcast_module(module, src=0)          # :176-179
    G_ema = copy.deepcopy(G).eval()
    stats = {}

    def report(name, value):
        stats.setdefault(name, []).append(value.detach().float().mean())
    lk = dict(loss_kwargs)
    loss = construct_class_by_name(device=device, G=G, D=D, augment_pipe=None, report_fn=report, **lk)
    phases = []
    for name, module, opt, reg in (('G', G, G_opt_kwargs, G_reg_interval), ('D', D, D_opt_kwargs, D_reg_interval)):
        opt = dict(opt)
        opt.pop('class_name', None)                   # torch.optim.Adam in the reference; the fused Adam kernel here
        phases.append(Phase(name + ('both' if reg is None else 'main'), module, lr=opt.get('lr', 1e-3), betas=tuple(opt.get('betas', (0.9, 0.999))),
                            eps=opt.get('eps', 1e-8), reg_interval=reg))
    dp = DataParallelStep(world_size=num_gpus)
    ema = EmaTracker(phases[0], G_ema)
    if rank == 0:
        print('Not run on this path: augment pipe, ADA, image/network snapshots, metrics (outside the hot path)')
        print(f'Training for {total_kimg} kimg...')
    cur_nimg, cur_tick, tick_start_nimg, tick_start = resume_kimg * 1000, 0, resume_kimg * 1000, time.time()
    if progress_fn is not None:
        progress_fn(0, total_kimg)
    last = {}
    while True:
        samples, real_c = next(it)
        texts = list(map(list, zip(*samples['texts']))) if isinstance(samples.get('texts'), (list, tuple)) else samples['texts']   # :246
        b = samples['bboxes'].shape[0]
        if isinstance(texts, list):     # strings -> tokens ONCE per iteration (the reference re-tokenises inside each of the 5 G/D forwards)
            from .networks_detr import _coerce_text
            texts = _coerce_text(G, texts, device)
        batch = dict(bbox_real=samples['bboxes'].to(device).float(), bbox_class=samples['labels'].to(device).long(), bbox_text=texts,
                     bbox_patch=samples['patches'].to(device), padding_mask=~samples['mask'].to(device).bool(),
                     background=samples['background'].to(device).float(), real_c=real_c.to(device))
        if batch['real_c'].shape[1] > 0 and hasattr(training_set, 'get_label'):      # training_loop.py:257-259: labels of random dataset items
            gen_c = np.stack([training_set.get_label(np.random.randint(len(training_set))) for _ in range(b)])
            batch['gen_c'] = torch.from_numpy(gen_c).to(device)
        else:
            batch['gen_c'] = torch.zeros_like(batch['real_c'])
        gen_z = [torch.randn(b, batch['bbox_class'].shape[1], G.z_dim, device=device) for _ in phases]
        training_iteration(loss, phases, dp, batch, batch_gpu, gen_z, ema=ema, batch_size=batch_size, ema_kimg=ema_kimg, cur_nimg=cur_nimg)
        cur_nimg += batch_size
        done = cur_nimg >= total_kimg * 1000
        if not done and cur_tick != 0 and cur_nimg < tick_start_nimg + kimg_per_tick * 1000:
            continue
        torch.cuda.synchronize()
        now = time.time()
        last = {k: torch.stack(v).mean().item() for k, v in stats.items()}
        stats.clear()
        if rank == 0:
            print(f'tick {cur_tick:<5d} kimg {cur_nimg / 1e3:<8.1f} sec/kimg {(now - tick_start) / max(cur_nimg - tick_start_nimg, 1) * 1e3:<7.2f} ' +
                  ' '.join(f'{k.split("/")[-1]} {v:.3f}' for k, v in sorted(last.items()) if k.startswith('Loss/scores')))
        if abort_fn is not None and abort_fn():
            done = True
        cur_tick += 1
        tick_start_nimg, tick_start = cur_nimg, time.time()
        if progress_fn is not None:
            progress_fn(cur_nimg // 1000, total_kimg)
        if done:
            break
    last['cur_nimg'] = cur_nimg
    return dict(stats=last, G=G, D=D, G_ema=G_ema)
