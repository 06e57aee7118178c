#!/usr/bin/env python
"""bench.py — LayoutDETR G+D adversarial step on MI355X (BASELINE.json metric).

A "step" is one full training iteration of the hot path: phase Gmain + phase Dmain
(G fwd x2, G bwd x1, D fwd x3, D bwd x3; training/loss.py:84-116,146-218 with gamma=0, pl_weight=0),
gradient exchange + /world + nan_to_num, Adam, G_ema lerp — in train mode (dropout 0.1), fp32.
Workload (BASELINE.json configs[2]/[3]): global batch 16, 256x256 synthetic backgrounds, 9 elements per layout,
hot-path-only (BASELINE.md variant A): the frozen BERT text encoder's CLS features are an input tensor and the
LM-decoder loss is excluded (SURVEY §8a rows a16/a17 are boundary inputs).
N > 1: one process per GPU.  `python bench.py --gpus N` spawns the N ranks itself (torch.multiprocessing.spawn, as the reference's
train.py:27-47 does); under torch.distributed.run it reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment instead.
The headline `value` at N > 1 is BASELINE configs[3]: GLOBAL batch 16 sharded over the ranks ("scaling": "strong"); the same
process then also times 16 samples PER GPU and reports it as `weak_scaling` (global batch 16 x N).  At N = 1 both coincide
(configs[2]) and the line also carries `value_reference_call_pattern` (D's trunk evaluated per D pass, as the reference does) and
`value_phase_trunk_sharing` (once per phase); the headline evaluates it once per iteration (same values: D's weights do not change
between Gmain and Dmain).
Gradients are exchanged with RCCL all-reduce over xGMI, bucketed and overlapped with the backward graphs.

Prints ONE JSON line on rank 0.
"""
import argparse
import copy
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

F32_MFMA_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense, 256 CUs @ 2.4 GHz
BF16_MFMA_PEAK_TFLOPS = 2500.0  # same guide: dense bf16 (v_mfma_f32_32x32x16_bf16); the exact 3-way split spends 6 bf16 products per fp32 product
SPLIT_PIPE_PEAK_TFLOPS = BF16_MFMA_PEAK_TFLOPS / 6.0   # fp32-equivalent ceiling of a launch that runs on the bf16 pipe with the split
METRIC = 'images/sec G+D fwd-bwd, 256x256 bg x9 elems'


def make_batch(b, bg, device, seed):
    g = torch.Generator().manual_seed(seed)
    xy = torch.rand(b, 9, 2, generator=g) * 0.6 + 0.2
    wh = torch.rand(b, 9, 2, generator=g) * 0.35 + 0.05
    bt = dict(bbox_real=torch.cat([xy, wh], -1), bbox_class=torch.randint(0, 8, (b, 9), generator=g),
              text_feat=torch.randn(b, 9, 768, generator=g), text_len=torch.randint(1, 40, (b, 9), generator=g),
              padding_mask=torch.zeros(b, 9, dtype=torch.bool), background=torch.randn(b, 3, bg, bg, generator=g))
    return bt


LAST_TEXT_EVAL = [None]     # token positions the last to_device_batch() kept per element text (None: text features in)


def to_device_batch(bt, device, text_mode='features', text_tokens=256, text_valid=(8, 40), text_padded=False):
    """text path on: synthetic tokenizer output as the reference's tokenizer produces it (networks_detr.py:71,145: padding='max_length', max_length 256) --
    random word ids on the first 8..40 positions, [PAD] = 0 behind them.  Unless text_padded, the all-padding tail shared by the whole
    batch is dropped on the host (what tokenizer.texts_to_tokens does for strings: T = the batch's longest text; same CLS features / LM loss / gradients,
    tests/test_composition_gpu.py::test_trimmed_text_tokens_equal_the_reference_padding_to_256) -> batch (LAST_TEXT_EVAL[0] = the evaluated token positions)."""
    from layoutdetr_amd.training.networks_detr import TextFeatures, TextTokens
    b = bt['bbox_real'].shape[0]
    t_eval = None
    if text_mode == 'features':
        text = TextFeatures(bt['text_feat'].to(device), bt['text_len'].to(device))
    else:
        g = torch.Generator().manual_seed(7)
        ids = torch.randint(1000, 30000, (b, 9, text_tokens), generator=g)
        lens = torch.randint(min(text_valid[0], text_tokens), min(text_valid[1], text_tokens) + 1, (b, 9), generator=g)
        am = (torch.arange(text_tokens)[None, None, :] < lens[..., None]).long()
        ids = ids * am
        t_eval = text_tokens if text_padded else max(int(lens.max()), 2)
        text = TextTokens(ids[..., :t_eval].contiguous().to(device), am[..., :t_eval].contiguous().to(device), bt['text_len'].to(device))
    batch = dict(bbox_real=bt['bbox_real'].to(device), bbox_class=bt['bbox_class'].to(device),
                 bbox_text=text,
                 bbox_patch=torch.zeros(b, 9, 1, 1, 1, device=device).expand(b, 9, 3, 256, 256),  # shape only (0-stride view)
                 padding_mask=bt['padding_mask'].to(device), background=bt['background'].to(device),
                 real_c=torch.zeros(b, 0, device=device), gen_c=torch.zeros(b, 0, device=device))
    LAST_TEXT_EVAL[0] = t_eval
    return batch


def _host_cpu():
    """(logical cores visible to this process, CPU model string) of the box."""
    nproc = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    model = 'unknown'
    try:
        with open('/proc/cpuinfo') as fh:
            for line in fh:
                if line.lower().startswith('model name'):
                    model = line.split(':', 1)[1].strip()
                    break
    except OSError:
        pass
    return nproc, model


def cpu_baseline(G_sd, D_sd, G_names, D_names, bg, budget_s=34.0):
    """The oracle (a CPU port of the same step) on the host cores, bounded sample: the headline's batch 16 -- one warm-up + up to three timed
    iterations (>= 3) as far as they fit the budget -- else batch 2 (and the unit says so).  `cores` = the threads used (the cap
    is measured, below), `host_cores` / `host_cpu` = what the box has."""
    from oracle import step_ref
    nproc, cpu_model = _host_cpu()
    ncores = min(nproc, 16)   # measured on the GPU box: 16 threads 1.1 s/iteration, 64 threads 3.3 s, 256 threads > 400 s (oversubscribed tiny ops)
    torch.set_num_threads(ncores)
    kw = dict(bg_size=bg, G_param_names=G_names, D_param_names=D_names)
    t_start = time.time()

    def leg(B, max_iters, deadline):
        bt = make_batch(B, bg, 'cpu', 123)
        zg, zd = torch.randn(B, 9, 4), torch.randn(B, 9, 4)
        t0 = time.time()
        step_ref.training_iteration(G_sd, D_sd, bt, zg, zd, **kw)   # warm-up (allocator, oneDNN primitive caches)
        warm = time.time() - t0
        if time.time() + warm > deadline:                            # a timed iteration would not fit
            return None, warm
        n, t0 = 0, time.time()
        while True:
            step_ref.training_iteration(G_sd, D_sd, bt, zg, zd, **kw)
            n += 1
            el = time.time() - t0
            if n >= max_iters or time.time() + el / n > deadline:
                break
        return (B * n / el, n), warm

    small, _ = leg(2, 3, t_start + 6.0)
    big, warm16 = leg(16, 3, t_start + budget_s)
    out = dict(cores=ncores, host_cores=nproc, host_cpu=cpu_model, kind='port',
               cores_note=f'{ncores} threads is the fastest setting of this port on the GPU boxes (measured: 16 threads 1.1 s / iteration at batch 16, 64 threads 3.3 s, all '
                          f'{nproc} logical cores > 400 s: ~2000 tiny ops per iteration oversubscribe the thread pool); all_cores = the same step at torch.set_num_threads(os.cpu_count()) '
                          '(BASELINE.md 2), batch 2, in a child process with a 25 s limit')
    out['all_cores'] = cpu_baseline_all_cores(bg, nproc)
    if big is not None:
        out.update(value=round(big[0], 4), unit='images/s', batch=16,
                   sample=f'{big[1]} timed iteration(s) of the same Gmain+Dmain step at batch 16, {bg}x{bg} (oracle/step_ref.py, torch CPU fp32, dropout off) after 1 warm-up')
        if small is not None:
            out['value_batch_2'] = round(small[0], 4)
    else:
        out.update(value=round(small[0], 4) if small is not None else None, unit='images/s at batch 2 (a batch-16 iteration did not fit the time budget)', batch=2,
                   sample=f'{small[1] if small else 0} timed iteration(s) of the same Gmain+Dmain step at batch 2, {bg}x{bg} (oracle/step_ref.py, torch CPU fp32, dropout off) after 1 warm-up; '
                          f'one batch-16 iteration took {warm16:.1f} s')
    return out


def cpu_baseline_all_cores(bg, nproc, limit_s=25.0):
    """BASELINE.md 2 asks for torch.set_num_threads(os.cpu_count()).  On the GPU boxes (256 logical cores) that setting is far slower than 16 threads and
    cannot be interrupted in-process, so it runs as a child (`bench.py --cpu-probe-threads N`) with a time limit: -> dict(threads, value | None, note)."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), '--cpu-probe-threads', str(nproc), '--bg', str(bg)], capture_output=True, text=True, timeout=limit_s,
                           env=dict(os.environ, HIP_VISIBLE_DEVICES='', CUDA_VISIBLE_DEVICES=''))
        line = [l for l in r.stdout.splitlines() if l.startswith('{')]
        if r.returncode == 0 and line:
            return json.loads(line[-1])
        return dict(threads=nproc, value=None, note=f'child failed (rc {r.returncode}): {r.stderr[-200:]}')
    except subprocess.TimeoutExpired:
        return dict(threads=nproc, value=None, unit='images/s at batch 2', note=f'one warm-up + one timed batch-2 iteration did not finish within {limit_s:.0f} s at {nproc} threads')


def cpu_probe(threads, bg):
    """Child of cpu_baseline_all_cores: the oracle step at batch 2 with `threads` threads, one warm-up + one timed iteration; prints one JSON line."""
    from layoutdetr_amd.training.networks_detr import Discriminator, Generator
    from oracle import step_ref
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    kw = dict(num_bbox_labels=8, img_channels=3, img_height=bg, img_width=bg, c_dim=0, background_size=bg, bert_f_dim=768, bert_num_heads=4, bert_num_encoder_layers=12,
              bert_num_decoder_layers=2, im_f_dim=512)
    G = Generator(z_dim=4, f_dim=256, num_heads=4, num_layers=8, text_mode='features', **kw)
    D = Discriminator(f_dim=256, num_heads=4, num_layers=8, text_mode='features', **kw)
    G_sd, D_sd = dict(G.state_dict()), dict(D.state_dict())
    names = dict(G_param_names={n for n, _ in G.named_parameters()}, D_param_names={n for n, _ in D.named_parameters()})
    bt = make_batch(2, bg, 'cpu', 123)
    zg, zd = torch.randn(2, 9, 4), torch.randn(2, 9, 4)
    step_ref.training_iteration(G_sd, D_sd, bt, zg, zd, bg_size=bg, **names)
    t0 = time.time()
    step_ref.training_iteration(G_sd, D_sd, bt, zg, zd, bg_size=bg, **names)
    el = time.time() - t0
    print(json.dumps(dict(threads=threads, value=round(2 / el, 4), unit='images/s at batch 2', sample='1 timed iteration of the same Gmain+Dmain step at batch 2 after 1 warm-up')), flush=True)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cpu-probe-threads', type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--per-gpu-batch', type=int, default=0, help='time ONLY this many samples per GPU (weak scaling: global batch = this x N)')
    ap.add_argument('--global-batch', type=int, default=0, help='time ONLY this GLOBAL batch (strong scaling); default: 16 = BASELINE configs[2]/[3]')
    ap.add_argument('--no-graph', action='store_true', help='eager launches instead of hipGraph replay of each phase')
    ap.add_argument('--bg', type=int, default=256)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-extra', action='store_true', help='skip the secondary measurements (reference call pattern at N=1, weak scaling at N>1)')
    ap.add_argument('--text-mode', default='features', choices=['features', 'encoder', 'encoder+lm'],
                    help="'features' (headline config: frozen-BERT CLS features are the input); 'encoder': token ids in, the frozen text encoder runs "
                         "inside every G/D forward; 'encoder+lm': plus the trainable LM text decoder and its loss (SURVEY 8f-1)")
    ap.add_argument('--text-tokens', type=int, default=256, help='token positions per element text as the tokenizer pads them (reference: max_text_length = 256, networks_detr.py:71,145)')
    ap.add_argument('--text-valid', default='8-40', help='range of real tokens per element text (SURVEY 8d: 8-40), the rest is [PAD]')
    ap.add_argument('--text-padded', action='store_true', help='evaluate all --text-tokens positions as the reference does, instead of the batch-longest text (same values, parity-tested)')
    ap.add_argument('--no-share-trunk', action='store_true', help="evaluate D's ResNet trunk separately for the fake and the real pass of Dmain, as the reference does")
    ap.add_argument('--share-trunk', default='iteration', choices=['phase', 'iteration'],
                    help="how often D's ResNet trunk runs on the iteration's backgrounds: 'iteration' (default) once -- D's weights do not change between Gmain and Dmain "
                         "(training_loop.py:281-313: Gmain steps G only), so Gmain's D(fake) reads the evaluation Dmain differentiates; 'phase': once per phase; "
                         "--no-share-trunk: once per D pass, the reference's call pattern.  Same values in all three (parity-tested); the line reports all three rates")
    ap.add_argument('--no-overlap', action='store_true', help='N>1: exchange gradients after each backward graph instead of overlapping the all-reduce with it')
    return ap.parse_args()


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


def _spawned(local_rank, args, port):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(local_rank), LOCAL_RANK=str(local_rank), WORLD_SIZE=str(args.gpus))
    run(args, local_rank, local_rank, args.gpus)


def main():
    args = parse_args()
    if args.cpu_probe_threads:
        return cpu_probe(args.cpu_probe_threads, args.bg)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # dmabuf IPC only on this host driver (RCCL needs it)
    env_world = int(os.environ.get('WORLD_SIZE', '0') or 0)
    if env_world >= 1 and 'RANK' in os.environ:            # launched by torch.distributed.run: one process per GPU already exists
        if env_world != args.gpus:
            raise SystemExit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={env_world}')
        # preflight: fail fast, with one parsable line, when this node shows fewer GPUs than local ranks -- never share a device silently
        # (RCCL refuses two ranks on one device only after every rank has entered init_process_group: minutes of rendezvous timeout)
        n_local = int(os.environ.get('LOCAL_WORLD_SIZE', env_world) or env_world)
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < n_local and not os.environ.get('LDETR_BENCH_SHARE_GPU'):
            if int(os.environ['RANK']) == 0:
                print(json.dumps(dict(metric=METRIC, value=None, unit='images/s', n_gpus=args.gpus, steps=args.steps, warmup=args.warmup,
                                      error=f'{n_local} local rank(s) launched but {have} GPU(s) visible on this node')), flush=True)
            raise SystemExit(2)
        run(args, int(os.environ['RANK']), int(os.environ.get('LOCAL_RANK', '0')), env_world)
    elif args.gpus > 1:                                      # plain `python bench.py --gpus N`: spawn the ranks (train.py:27-47 does the same)
        import torch.multiprocessing as mp
        if not torch.cuda.is_available() or not (torch.cuda.device_count() >= args.gpus or os.environ.get('LDETR_BENCH_SHARE_GPU')):
            # one parsable line instead of a traceback: the driver's scaling sweep asks for N = 1, 2, 4, 8 on whatever node it got
            print(json.dumps(dict(metric=METRIC, value=None, unit='images/s', n_gpus=args.gpus, steps=args.steps, warmup=args.warmup,
                                  error=f'--gpus {args.gpus} requested but {torch.cuda.device_count() if torch.cuda.is_available() else 0} GPU(s) visible on this node')), flush=True)
            raise SystemExit(2)
        mp.spawn(_spawned, args=(args, _free_port()), nprocs=args.gpus, join=True)
    else:
        run(args, 0, 0, 1)


def run(args, rank, local_rank, world):
    assert torch.cuda.is_available(), 'bench.py needs a GPU'
    # development aid (a 1-GPU box): LDETR_BENCH_SHARE_GPU=1 puts every rank on cuda:0 and exchanges through gloo (RCCL refuses two ranks
    # on one device); it exercises the rank logic, the staged graphs and the overlap plumbing, not xGMI
    share_gpu = bool(os.environ.get('LDETR_BENCH_SHARE_GPU'))
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if share_gpu:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=device)
        assert dist.get_world_size() == world

    from layoutdetr_amd import _lib
    _lib.load()   # fails loudly if the HIP library is missing: there is no fallback path
    from layoutdetr_amd.hip import core
    from layoutdetr_amd.training import training_loop as tl
    from layoutdetr_amd.training.loss import StyleGAN2Loss
    from layoutdetr_amd.training.networks_detr import Discriminator, Generator

    bg = args.bg
    torch.manual_seed(0)   # identical initial parameters on every rank (stands in for the rank-0 broadcast, training_loop.py:176-179)
    kw = dict(num_bbox_labels=8, img_channels=3, img_height=bg, img_width=bg, c_dim=0, background_size=bg, bert_f_dim=768,
              bert_num_heads=4, bert_num_encoder_layers=12, bert_num_decoder_layers=2, im_f_dim=512)
    G = Generator(z_dim=4, f_dim=256, num_heads=4, num_layers=8, text_mode=args.text_mode, **kw).train().requires_grad_(False)
    D = Discriminator(f_dim=256, num_heads=4, num_layers=8, text_mode=args.text_mode, **kw).train().requires_grad_(False)
    G_sd_cpu = D_sd_cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.text_mode == 'features':
        G_sd_cpu = {k: v.clone() for k, v in G.state_dict().items()}
        D_sd_cpu = {k: v.clone() for k, v in D.state_dict().items()}
    G_names = {n for n, _ in G.named_parameters()}
    D_names = {n for n, _ in D.named_parameters()}
    G.to(device); D.to(device)
    G.static_shapes = D.static_shapes = True   # sync-free heads/losses (same values; required for graph capture)
    G_ema = copy.deepcopy(G).eval()
    pG = tl.Phase('Gmain', G, lr=1e-5, betas=(0.0, 0.99), eps=1e-8, reg_interval=4)     # train.py:204,281; training_loop.py:191-194
    pD = tl.Phase('Dmain', D, lr=1e-5, betas=(0.0, 0.99), eps=1e-8, reg_interval=16)
    ema = tl.EmaTracker(pG, G_ema)
    dp_world = tl.DataParallelStep(world_size=world)
    n_params = (pG.fm.total, pD.fm.total)
    torch.manual_seed(0 * world + rank)   # training_loop.py:101-102 seed rule
    side = torch.cuda.Stream()
    text_eval = [None]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(b_local, share, want_eager=False, local_only=False):
        """W warm-up + K timed iterations at `b_local` samples per GPU; -> dict(value, ms_per_step, global_batch[, eager_step]).
        local_only (N > 1 diagnostics): the same step WITHOUT the gradient exchange (every rank steps on its own, as at N = 1): what the
        driver's N = 1 bench line measures, re-measured on this node's GPUs."""
        gb = b_local * world
        dp = dp_world if not local_only else tl.DataParallelStep(world_size=1)
        loss = StyleGAN2Loss(device, G, D, share_D_trunk=share)
        lo_, _, hi_ = args.text_valid.partition('-')
        batch = to_device_batch(make_batch(b_local, bg, device, 1000 + rank), device, args.text_mode, args.text_tokens, (int(lo_), int(hi_ or lo_)), args.text_padded)
        text_eval[0] = LAST_TEXT_EVAL[0]
        cur_nimg = [0]

        def eager_step():
            gen_z = [torch.randn(b_local, 9, 4, device=device) for _ in range(2)]
            tl.training_iteration(loss, [pG, pD], dp, batch, b_local, gen_z, ema=ema, batch_size=gb, ema_kimg=gb * 10 / 32, cur_nimg=cur_nimg[0])
            cur_nimg[0] += gb

        step, stage_info = eager_step, {}
        if not args.no_graph:
            # eager warm-up before capture (allocator, folded-BN / position-encoding caches) on a SIDE stream: autograd's AccumulateGrad
            # nodes remember the stream they were first used on, and one bound to the default stream breaks a later capture
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    eager_step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graphed = tl.GraphedIteration(loss, [pG, pD], dp, batch, b_local, 4, ema=ema, batch_size=gb, ema_kimg=gb * 10 / 32,
                                          capture_stream=side, overlap=(dp.world > 1 and not args.no_overlap))
            graphed.cur_nimg = cur_nimg[0]
            step = graphed.run
            stage_info = dict(backward_stages=graphed.n_stages, backward_stage_ms=[round(v, 2) for v in graphed.stage_ms] if graphed.stage_ms else None)
        for _ in range(args.warmup):
            step()
        dp.exposed.clear()
        dp.record_exposed = world > 1 and not local_only
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        mine = time.perf_counter() - t0          # this rank's own clock, before the closing barrier
        barrier()
        elapsed = time.perf_counter() - t0
        dp.record_exposed = False
        diag = None
        if world > 1:
            t = torch.tensor([elapsed], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = t.item()
            # self-diagnosis of a multi-GPU run: per-rank step time (a straggler shows as spread) and, per phase, how long the compute stream
            # stood still waiting for the all-reduce (event pair around DataParallelStep's join): the part of the exchange NOT hidden behind backward
            exp = {}
            for name, e0, e1 in dp.exposed:
                exp[name] = exp.get(name, 0.0) + e0.elapsed_time(e1)
            names = sorted({p.name for p in (pG, pD)})
            vec = torch.tensor([mine / args.steps * 1e3] + [exp.get(n, 0.0) / args.steps for n in names], device=device, dtype=torch.float64)
            allv = [torch.zeros_like(vec) for _ in range(world)]
            dist.all_gather(allv, vec)
            allv = torch.stack(allv).cpu()
            diag = dict(rank_ms_per_step=dict(min=round(allv[:, 0].min().item(), 3), max=round(allv[:, 0].max().item(), 3), mean=round(allv[:, 0].mean().item(), 3),
                                              per_rank=[round(v, 3) for v in allv[:, 0].tolist()]))
            if not local_only:
                diag['comm_exposed_ms'] = {n: dict(mean=round(allv[:, 1 + i].mean().item(), 3), max=round(allv[:, 1 + i].max().item(), 3)) for i, n in enumerate(names)}
                diag['comm_exposed_ms']['per_step_total_max'] = round(allv[:, 1:].sum(1).max().item(), 3)
                diag['allreduce_mb_per_step'] = round(4e-6 * (pG.fm.total + pD.fm.total), 1)
        out = dict(value=round(gb * args.steps / elapsed, 3), ms_per_step=round(elapsed / args.steps * 1e3, 3), global_batch=gb, per_gpu_batch=b_local)
        if diag is not None:
            diag.update(stage_info)      # stages of the overlapped backward, chosen from the measured stage lengths (training_loop.backward_stage_count)
            out['diagnostics'] = diag
        if want_eager:
            out['eager_step'] = eager_step
        return out

    share = False if args.no_share_trunk else (True if args.share_trunk == 'phase' else 'iteration')
    extra = {}
    if args.per_gpu_batch:
        primary, scaling = measure(args.per_gpu_batch, share, want_eager=True), 'weak'
    else:
        gbatch = args.global_batch or 16
        assert gbatch % world == 0, f'global batch {gbatch} does not divide over {world} GPUs'
        primary, scaling = measure(gbatch // world, share, want_eager=True), ('strong' if world > 1 else 'weak')
        if not args.no_extra and not args.global_batch:
            if world > 1:      # the same ranks at 16 samples per GPU (weak scaling: global batch 16 x N)
                w = measure(16, share)
                extra['weak_scaling'] = dict(value=w['value'], ms_per_step=w['ms_per_step'], global_batch=w['global_batch'], per_gpu_batch=16, unit='images/s',
                                             diagnostics=w.get('diagnostics'))
                # ... and WITHOUT the exchange: every rank runs the N = 1 bench workload on its own GPU.  `value_per_gpu` is what BENCH (N = 1)
                # reports, re-measured here: it must agree with the driver's N = 1 line (else this node / these GPUs differ), and
                # weak_scaling.ms_per_step - this ms_per_step is the whole cost of data parallelism (exposed exchange + stragglers).
                l = measure(16, share, local_only=True)
                extra['single_gpu_reference'] = dict(value_per_gpu=round(l['value'] / world, 3), ms_per_step=l['ms_per_step'], per_gpu_batch=16, unit='images/s',
                                                     note='same processes, 16 samples per GPU, no gradient exchange (= the N=1 bench workload on each GPU); compare with BENCH at N=1',
                                                     rank_ms_per_step=(l.get('diagnostics') or {}).get('rank_ms_per_step'))
                # what strong scaling of the 16-sample step can reach before any communication: the step at 16 / N samples per GPU is bound by its
                # launch-latency floor, not by its FLOPs
                ls = measure(gbatch // world, share, local_only=True)
                extra['strong_scaling_ceiling'] = dict(value=round(l['ms_per_step'] / ls['ms_per_step'], 3), n_gpus=world,
                                                       ms_per_step_16_per_gpu=l['ms_per_step'], ms_per_step_share=ls['ms_per_step'], per_gpu_batch_share=gbatch // world,
                                                       note='ms_per_step(16 samples on one GPU) / ms_per_step(16/N samples on one GPU), both without gradient exchange: the '
                                                            'speed-up of the headline (global batch 16) over N = 1 cannot exceed this; weak_scaling is the figure that scales with N')
            elif share is not False:   # N = 1: the reference's call pattern (one D-trunk evaluation per D pass: 25 % more conv FLOPs per step)
                r = measure(16, False)
                extra['value_reference_call_pattern'] = r['value']
                extra['ms_per_step_reference_call_pattern'] = r['ms_per_step']
                # ... and the setting in between / beyond the headline's: D's trunk once per phase, once per iteration
                if share == 'iteration':
                    extra['value_phase_trunk_sharing'] = measure(16, True)['value']
                else:
                    extra['value_iteration_trunk_sharing'] = measure(16, 'iteration')['value']
                # the same step with every contraction on the f32 MFMA pipe (the default runs the 128-row / narrow tiles of the engine on the
                # bf16 pipe with the exact three-way operand split: fp32 operands and results, csrc/gemm_conv.hip gemm_f32_kernel<.., SPLIT>)
                prev = core.lib().ldetr_set_split_bf16(0)
                prev_p3 = os.environ.get('LDETR_TRUNK_P3')
                os.environ['LDETR_TRUNK_P3'] = '0'
                try:
                    r = measure(16, share)
                finally:
                    core.lib().ldetr_set_split_bf16(prev)
                    if prev_p3 is None:
                        os.environ.pop('LDETR_TRUNK_P3', None)
                    else:
                        os.environ['LDETR_TRUNK_P3'] = prev_p3
                    tl.refresh_weight_planes(G); tl.refresh_weight_planes(D)   # the weights moved while the plane-format trunk was off
                extra['value_f32_mfma_only'] = r['value']
                # the bound SCALE will hit at 8 GPUs (BASELINE configs[3] = the reference's recommended launch: global batch 16 -> 2 samples per GPU):
                # a 2-sample step is a launch-latency chain, so 16 samples on one GPU / 2 samples on one GPU is all strong scaling can return
                s2 = measure(2, share)
                extra['strong_scaling_ceiling'] = dict(value=round(primary['ms_per_step'] / s2['ms_per_step'], 3), n_gpus=8, ms_per_step_16_per_gpu=primary['ms_per_step'],
                                                       ms_per_step_share=s2['ms_per_step'], per_gpu_batch_share=2,
                                                       note='ms_per_step(16 samples on one GPU) / ms_per_step(2 samples on one GPU), no gradient exchange: the speed-up of a global batch of 16 '
                                                            'on 8 GPUs over N = 1 cannot exceed this (measured here on ONE GPU); weak scaling (16 per GPU) is the figure that scales with N')
    eager_step = primary.pop('eager_step')
    if primary.get('diagnostics') is not None:
        extra['diagnostics'] = primary.pop('diagnostics')
    args.batch, b_local = primary['global_batch'], primary['per_gpu_batch']
    value, ms_per_step = primary['value'], primary['ms_per_step']

    roofline = None
    if not args.no_roofline:
        # Dominant kernel = the f32-MFMA contraction engine (ldetr::gemm_f32_kernel<...>, every dense GEMM and implicit conv).
        # HIP events on the launch stream around each engine launch, algorithmic FLOPs = 2*M*N*K (GEMM) /
        # 2*pixels*Cout*KH*KW*Cin (conv fwd, bwd-data, bwd-weight alike), over 2 extra iterations.
        # Per-launch events need eager launches (same kernels, same shapes as the replayed graphs).  The CPU enqueues small
        # launches slower than the GPU retires them, and an idle gap between the start event and the kernel would be billed
        # to the kernel; so the stream is first held busy by a calibrated spin kernel while the CPU runs a whole step ahead.
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record(); torch.cuda._sleep(20_000_000); ev1.record(); torch.cuda.synchronize()
        cycles_per_ms = 20_000_000 / max(ev0.elapsed_time(ev1), 1e-3)
        core.PROF.enabled = True
        core.PROF.reset()
        for _ in range(2):
            torch.cuda.synchronize()
            torch.cuda._sleep(int(cycles_per_ms * 2.0 * ms_per_step))
            eager_step()
        torch.cuda.synchronize()
        fl, sec_raw, launches = core.PROF.summary()
        core.PROF.enabled = False
        # An event pair has a cost of its own (marker packets before and after the launch).  Calibration: the same small engine launch
        # N times inside ONE event span vs N times with a pair around each; the difference per launch is what a pair adds, and it is
        # subtracted per engine launch.  With it the summed time agrees with rocprofv3's kernel durations (profiles/r01g: 51.0 ms).
        ca, cw = torch.randn(64, 64, device=device), torch.randn(64, 64, device=device)
        cy = torch.empty(64, 64, device=device)
        tiny = lambda: core.gemm(ca, cw, 0, 0, 64, 64, 64, out=cy)
        ncal = 256
        tiny(); torch.cuda.synchronize()
        torch.cuda._sleep(int(cycles_per_ms * 20.0))
        sa, sb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        sa.record()
        for _ in range(ncal):
            tiny()
        sb.record()
        torch.cuda._sleep(int(cycles_per_ms * 20.0))
        pairs = []
        for _ in range(ncal):
            a_ = torch.cuda.Event(enable_timing=True); b_ = torch.cuda.Event(enable_timing=True)
            a_.record(); tiny(); b_.record(); pairs.append((a_, b_))
        torch.cuda.synchronize()
        ev_over_ms = max((sum(x.elapsed_time(y) for x, y in pairs) - sa.elapsed_time(sb)) / ncal, 0.0)
        sec = max(sec_raw - ev_over_ms * 1e-3 * launches, 1e-9)
        # per (kernel symbol, entry point, shape) table of the step's contraction launches: calls, mean duration, algorithmic FLOPs and bytes per
        # launch -- what a single kernel's roofline is recomputed from (LDETR_ENGINE_SHAPES=<file> writes it: profiles/r06_engine_shapes.txt)
        shapes = {}
        for tag, f, s_, e_, nb, pp, lab in core.PROF.records:
            a = shapes.setdefault((lab, tag, f, nb), [0, 0.0, pp]); a[0] += 1; a[1] += s_.elapsed_time(e_)
        if os.environ.get('LDETR_ENGINE_SHAPES'):
            with open(os.environ['LDETR_ENGINE_SHAPES'], 'w') as fh:
                fh.write(f'# {METRIC}; {b_local} samples per GPU, {bg}x{bg}; one row per (kernel symbol, C-ABI entry, algorithmic FLOPs, algorithmic bytes) of ONE iteration; '
                         'us = mean HIP-event time of a call (event pair included); pipe = matrix pipe of its launches\n')
                fh.write(f'# {"kernel":58s} {"entry":34s} {"calls/step":>10s} {"us/call":>9s} {"GFLOP/call":>11s} {"MB/call":>9s} {"TFLOP/s":>8s} {"TB/s":>6s}  pipe\n')
                for (lab, tag, f, nb), (n, ms, pp) in sorted(shapes.items(), key=lambda kv: -kv[1][1]):
                    fh.write(f'{lab:60s} {tag.replace("ldetr_", ""):34s} {n / 2:10.1f} {ms / n * 1e3:9.1f} {f / 1e9:11.4f} {nb / 1e6:9.2f} {f * n / ms / 1e9 if ms else 0:8.1f} '
                             f'{nb * n / ms / 1e9 if ms else 0:6.2f}  {"bf16x6" if pp[1] else "f32"}\n')
        # `achieved` / `frac` use the UNCORRECTED event time: it is the figure that agrees with rocprofv3's kernel durations of the same
        # command (round 3: 32.06 ms of events against 32.0 ms in profiles/r03g_kernel_stats.csv); the pair-corrected one is reported beside it
        ach = fl / sec_raw / 1e12 if sec_raw > 0 else 0.0
        ach_corr = fl / sec / 1e12 if sec > 0 else 0.0
        by = {}
        t_pipe = [0.0, 0.0]          # engine time on the f32 MFMA pipe / on the bf16 pipe (exact operand split)
        n_pipe = [0, 0]
        alg_bytes = alg_bytes_f32 = 0.0
        for tag, f, s_, e_, nb, pipes, _lab in core.PROF.records:   # the same records, split by C-ABI entry point
            key = 'dense_gemm' if tag == 'gemm' else tag.replace('ldetr_', '').replace('_f32', '')
            ms = s_.elapsed_time(e_)
            a = by.setdefault(key, [0.0, 0.0, 0, 0.0, 0.0]); a[0] += f; a[1] += ms; a[2] += 1; a[3] += nb
            alg_bytes += nb
            alg_bytes_f32 += nb * (4.0 / 6.0 if tag.startswith('ldetr_p3') else 1.0)      # plane-format operands are 6 bytes per element: SURVEY 8d counts fp32 tensors
            tot = max(pipes[0] + pipes[1], 1)
            t_pipe[0] += ms * pipes[0] / tot; t_pipe[1] += ms * pipes[1] / tot      # (a call that issued launches on both pipes: split by launch count)
            n_pipe[0] += pipes[0]; n_pipe[1] += pipes[1]
            a[4] += ms * pipes[1] / tot
        by_entry = {k: dict(gflop_per_step=round(v[0] / 2 / 1e9, 1), ms_per_step=round(v[1] / 2, 2), launches_per_step=v[2] // 2,
                            tflops=round(v[0] / v[1] / 1e9, 2) if v[1] > 0 else 0.0, algorithmic_gb_per_step=round(v[3] / 2 / 1e9, 2),
                            frac_time_on_bf16_pipe=round(v[4] / v[1], 3) if v[1] > 0 else 0.0) for k, v in sorted(by.items())}
        # the ceiling in use: a launch on the bf16 pipe with the exact 3-way split can reach 2500 / 6 = 416.7 fp32-equivalent TFLOP/s, a
        # launch on the f32 MFMA pipe 157.3 -- time-weighted over the step's engine launches
        tsum = max(t_pipe[0] + t_pipe[1], 1e-9)
        peak_eff = (t_pipe[0] * F32_MFMA_PEAK_TFLOPS + t_pipe[1] * SPLIT_PIPE_PEAK_TFLOPS) / tsum
        engine = dict(bound='mfma', kernel='fp32-equivalent contraction engine, every launch: ldetr::p3_nt_kernel<*> / p3_c3_kernel / p3_tn_kernel<*> (the ResNet trunk on plane-format '
                                             'operands, bf16 pipe) + gemm_f32_kernel<*> (LDS-tiled GEMM / implicit conv) + gemm_small_kernel<*> / gemm_small_pair_kernel<*> '
                                             '(+ conv3x3_c32 / wgrad_c32 where they replace engine launches) + the token-stack kernels (mha_small / mha_cross / ffn fwd + bwd, wgrad_multi: f32 MFMA)',
                        achieved=round(ach, 3), peak=F32_MFMA_PEAK_TFLOPS, unit='TFLOP/s', frac=round(ach / F32_MFMA_PEAK_TFLOPS, 4),
                        achieved_event_corrected=round(ach_corr, 3), frac_event_corrected=round(ach_corr / F32_MFMA_PEAK_TFLOPS, 4),
                        peak_effective=round(peak_eff, 1), frac_effective=round(ach / peak_eff, 4),
                        peak_effective_note=f'time-weighted over the engine launches of the step: {t_pipe[1] / tsum:.3f} of the engine time runs on the bf16 matrix pipe with the '
                                            f'exact 3-way operand split (fp32-equivalent ceiling {SPLIT_PIPE_PEAK_TFLOPS:.1f} = 2500 / 6 TFLOP/s; {n_pipe[1] // 2} kernel launches per step), '
                                            f'{t_pipe[0] / tsum:.3f} on the f32 MFMA pipe ({F32_MFMA_PEAK_TFLOPS}; {n_pipe[0] // 2} launches)',
                        engine_ms_on_bf16_pipe=round(t_pipe[1] / 2, 3), engine_ms_on_f32_pipe=round(t_pipe[0] / 2, 3),
                        launches_per_step=launches // 2, algorithmic_gflop_per_step=round(fl / 2 / 1e9, 2),
                        algorithmic_gb_per_step=round(alg_bytes / 2 / 1e9, 2), algorithmic_gb_per_step_fp32_tensors=round(alg_bytes_f32 / 2 / 1e9, 2),
                        engine_ms_per_step=round(sec / 2 * 1e3, 3), engine_ms_per_step_uncorrected=round(sec_raw / 2 * 1e3, 3),
                        event_pair_overhead_us=round(ev_over_ms * 1e3, 2), by_entry=by_entry,
                        matrix_pipe='fp32 values, fp32 accumulators and results throughout; the ResNet trunk keeps its activations and weights as the exact 3-way bf16 split '
                                    '(plane format, csrc/p3_engine.hip: no conversion in the k-loop) and the 128x128 / 128x64 / 256x32 tiles of gemm_f32_kernel split their fp32 operands '
                                    'on the fly; both multiply on the bf16 pipe (6 x v_mfma_f32_32x32x16_bf16 per k16, error <= the f32 MFMA path: tests/test_p3_gpu.py, '
                                    'tests/test_kernels_gpu.py test_split_bf16_*), every other launch on v_mfma_f32_32x32x2_f32 / 16x16x4_f32; `peak` stays the f32 MFMA peak, `achieved` counts '
                                    'algorithmic fp32 FLOPs (a split launch can exceed it: 6/16 of the bf16 pipe time per fp32 FLOP); value_f32_mfma_only = same step with every split off '
                                    '(LDETR_TRUNK_P3=0, ldetr_set_split_bf16(0))')
        # ---- the dominant kernel: ONE kernel symbol (the one with the largest summed time in the step), priced against the pipe it runs on
        ksym = {}
        for tag, f, s_, e_, nb, pp, lab in core.PROF.records:
            if tag == 'ldetr_token_stack':
                continue                                   # a group of kernels behind one label, not a symbol
            a = ksym.setdefault(lab.split(' splitK')[0], dict(flops=0.0, ms=0.0, n=0, nbytes=0.0, split=0))
            a['flops'] += f; a['ms'] += s_.elapsed_time(e_); a['n'] += 1; a['nbytes'] += nb; a['split'] += 1 if pp[1] else 0

        def kernel_entry(sym):
            a = ksym[sym]
            on_bf16 = a['split'] * 2 > a['n']
            peak = SPLIT_PIPE_PEAK_TFLOPS if on_bf16 else F32_MFMA_PEAK_TFLOPS
            tf = a['flops'] / a['ms'] / 1e9 if a['ms'] > 0 else 0.0
            return dict(kernel=sym, launches_per_step=a['n'] // 2, avg_us=round(a['ms'] / a['n'] * 1e3, 2), ms_per_step=round(a['ms'] / 2, 3),
                        gflop_per_launch=round(a['flops'] / a['n'] / 1e9, 4), algorithmic_mb_per_launch=round(a['nbytes'] / a['n'] / 1e6, 3),
                        achieved=round(tf, 2), peak=round(peak, 1), unit='TFLOP/s', frac=round(tf / peak, 4), frac_of_f32_mfma_peak=round(tf / F32_MFMA_PEAK_TFLOPS, 4),
                        pipe='bf16 matrix pipe, exact 3-way operand split (6 products per fp32 product: 2500 / 6)' if on_bf16 else 'f32 MFMA',
                        algorithmic_tbps=round(a['nbytes'] / a['ms'] / 1e9, 3) if a['ms'] > 0 else 0.0)
        order = sorted(ksym, key=lambda k: -ksym[k]['ms'])
        roofline = dict(bound='mfma', **kernel_entry(order[0]), traffic=None,
                        definition='achieved = algorithmic FLOPs of this kernel\'s launches in the step (2 x pixels x Cout x KH x KW x Cin per convolution, data + weight gradient '
                                   'for a paired launch) / their summed HIP-event time, measured live on the launch stream; peak = the matrix pipe the kernel issues on; per-shape rows: '
                                   'profiles/r06_engine_shapes.txt; rocprofv3 durations of the same command: profiles/r06*_kernel_stats.csv',
                        by_kernel=[kernel_entry(k) for k in order[:12]], engine=engine)
        # HBM traffic of the engine from the PMC counters (FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3 passes over the same step, eager):
        # measured offline with tools/pmc_step.py (rocprofv3 cannot wrap this process from inside) and committed; per launch, like `achieved`
        pmc_path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
        if os.path.exists(pmc_path) and args.bg == 256 and b_local == 16 and args.text_mode == 'features':
            from layoutdetr_amd import build as kbuild
            pmc = json.load(open(pmc_path))
            if pmc.get('csrc_digest') != kbuild.source_digest():
                # counters of another build of the kernels are not this build's traffic: say so instead of quoting them
                roofline['traffic_note'] = engine['traffic_note'] = (f"profiles/pmc_traffic.json was measured on kernel sources {str(pmc.get('csrc_digest'))[:12]}, this build is "
                                            f"{kbuild.source_digest()[:12]}: re-run tools/pmc_step.py (two rocprofv3 --pmc passes) to refresh it")
            else:
                engine['traffic'] = round(pmc['engine_bytes_per_launch'])
                engine['traffic_unit'] = 'HBM bytes per engine launch (mean over the step)'
                engine['traffic_gb_per_step'] = round((pmc['engine_total']['fetch'] + pmc['engine_total']['write']) / 1e9, 2)
                engine['traffic_over_algorithmic'] = round(engine['traffic_gb_per_step'] / max(engine['algorithmic_gb_per_step'], 1e-9), 2)
                engine['traffic_over_fp32_algorithmic'] = round(engine['traffic_gb_per_step'] / max(engine['algorithmic_gb_per_step_fp32_tensors'], 1e-9), 2)
                src = f"profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, two passes over tools/pmc_step.py; kernel sources {pmc['csrc_digest'][:12]} = this build)"
                engine['traffic_source'] = src

                def pmc_of(sym):
                    """HBM bytes per launch of kernel symbol `sym` (bench label, e.g. 'p3_bwd_pair_nt_kernel<64,64,4w>' or 'gemm_f32_kernel<128,64,32,4w,FAST,SPLIT>') from
                    rocprofv3's per-instantiation counters ('p3_bwd_pair_nt_kernel<64, 64, 1, true>', 'gemm_f32_kernel<128, 64, 32, 2, 6, 4, true, true>'): same kernel name, same
                    tile (the leading two / three integers), same FAST / SPLIT flags where the label carries them; several operand-mode instantiations of one tile are summed."""
                    import re
                    base = sym.split('<')[0]
                    args = sym.split('<', 1)[1].rstrip('>') if '<' in sym else ''
                    nums = re.findall(r'\d+', args)[:3 if base == 'gemm_f32_kernel' else 2]
                    flags = None
                    if base == 'gemm_f32_kernel':
                        flags = ['true' if 'FAST' in args else 'false', 'true' if 'SPLIT' in args else 'false']
                    hits = []
                    for k, v in pmc['by_kernel'].items():
                        if k.split('<')[0] != base:
                            continue
                        kargs = [a.strip() for a in (k.split('<', 1)[1].rstrip('>').split(',') if '<' in k else [])]
                        if [a for a in kargs if a.isdigit()][:len(nums)] != nums:
                            continue
                        if flags is not None and kargs[-2:] != flags:
                            continue
                        hits.append(v)
                    if not hits:
                        return None
                    return sum(h['fetch'] + h['write'] for h in hits) / max(sum(h['launches'] for h in hits), 1)
                for ent in [roofline] + roofline['by_kernel']:
                    t = pmc_of(ent['kernel'])
                    if t is not None:
                        ent['traffic'] = round(t)
                        ent['traffic_over_algorithmic'] = round(t / max(ent['algorithmic_mb_per_launch'] * 1e6, 1.0), 2)
                roofline['traffic_unit'] = 'HBM bytes per launch of this kernel (FETCH_SIZE x 2 + WRITE_SIZE, mean over its launches in the step)'
                roofline['traffic_source'] = src
        try:    # the fractions north_star names, each as its own entry (HBM-bound kernels, modulated-conv layer at 256x256, DETR cross-attention)
            sys.path.insert(0, os.path.join(ROOT, 'tools'))
            import bench_hbm_kernels
            roofline['kernels'] = bench_hbm_kernels.measure_named(16)
        except Exception as err:   # a reporting leg: never fail the bench line for it
            roofline['kernels'] = f'unavailable: {err!r}'

    if rank == 0:
        cpu = None
        if G_sd_cpu is not None:
            try:
                cpu = cpu_baseline(G_sd_cpu, D_sd_cpu, G_names, D_names, bg)
            except Exception as err:   # a reporting leg: never fail the bench line for it
                cpu = dict(value=None, unit='images/s', cores=0, kind='port', sample=f'unavailable: {err!r}')
        out = dict(metric=METRIC, value=round(value, 3), unit='images/s', n_gpus=world, rccl_ranks=(dist.get_world_size() if world > 1 else 1), **({'shared_single_gpu_gloo': True} if share_gpu else {}),
                   steps=args.steps, warmup=args.warmup, ms_per_step=round(ms_per_step, 3), higher_is_better=True,
                   scaling=scaling, vs_baseline=None, dtype='f32', data='synthetic',
                   config=dict(reference_call_pattern_images_s=extra.get('value_reference_call_pattern'), phase_trunk_sharing_images_s=extra.get('value_phase_trunk_sharing'),
                               headline_note="value = D's ResNet trunk evaluated once per iteration (value-identical: D's weights do not move between Gmain and Dmain); "
                                             'quote it with phase_trunk_sharing_images_s (once per phase) and reference_call_pattern_images_s (once per D pass, as training/loss.py does)',
                               workload=f'BASELINE configs[{2 if world == 1 else 3}]: global batch {args.batch} ({b_local} per GPU), {bg}x{bg} backgrounds x 9 elements, full G+D adversarial '
                                        'step (Gmain+Dmain fwd/bwd, grad exchange + nan_to_num, Adam, EMA), train mode (dropout 0.1); ' +
                                        ('hot-path-only: frozen-BERT text features are an input, LM-decoder loss excluded' if args.text_mode == 'features' else f'text path on: {args.text_mode}, element texts of {args.text_valid} tokens padded to {args.text_tokens} as the reference tokenizer does; ' + (f'all {args.text_tokens} positions evaluated' if args.text_padded else f'the batch-longest text ({text_eval[0]} positions) evaluated: same values, parity-tested')),
                               parity_notes='values checked against oracle/ (pinned to the reference by tests/golden/*; the torchvision ResNet-50 body of the oracle is parity-UNPINNED: torchvision is '
                                            'absent from the build container, SURVEY 8c); dropout (on in this timed step, as in the reference) is validated statistically only; D-trunk sharing is value-identical '
                                            '(tests/test_model_gpu.py::test_iteration_level_D_trunk_sharing_matches_reference_call_pattern)',
                               global_batch=args.batch, per_gpu_batch=b_local, background=bg, elements=9,
                               parallelism=f'dp{world}', hip_graph=not args.no_graph, allreduce_overlapped_with_backward=(world > 1 and not args.no_graph and not args.no_overlap), text_mode=args.text_mode, text_tokens=(args.text_tokens if args.text_mode != 'features' else None), text_tokens_evaluated=text_eval[0], d_trunk_shared=False if args.no_share_trunk else args.share_trunk, params_G=n_params[0], params_D=n_params[1]),
                   roofline=roofline, cpu_baseline=cpu, **extra)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
