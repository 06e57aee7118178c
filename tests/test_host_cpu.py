"""CPU tests (no GPU): the C-ABI library loads and exports every symbol include/ldetr_hip.h declares, the host
logic around the kernels (flat parameter buffers, DP gradient exchange over gloo world_size 2, torch-glue losses
and position encoding against reference golden vectors), and that the product path refuses CPU tensors."""
import os
import re
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, 'tests', 'golden')


def load(name):
    d = np.load(os.path.join(G, name + '.npz'), allow_pickle=False)
    return {k: torch.from_numpy(np.asarray(d[k])) for k in d.files if d[k].dtype.kind in 'fiub'}


def test_library_exports_every_declared_symbol():
    from layoutdetr_amd import _lib
    hdr = open(os.path.join(ROOT, 'include', 'ldetr_hip.h')).read()
    declared = set(re.findall(r'^\s*(?:int|const char\*)\s+(ldetr_\w+)\s*\(', hdr, flags=re.M))
    assert len(declared) >= 25
    lib = _lib.load()                      # builds nothing: the .so must already exist (build() made it)
    for name in declared:
        assert hasattr(lib, name), f'{name} declared in include/ldetr_hip.h but not exported'
    assert declared - {'ldetr_last_error', 'ldetr_abi_version'} == set(_lib.SIGNATURES), 'ctypes table out of sync with the header'
    assert lib.ldetr_abi_version() == _lib.ABI_VERSION == 24


def test_no_cpu_fallback():
    from layoutdetr_amd.hip import conv, core, linear
    from layoutdetr_amd.torch_utils.ops import bias_act, upfirdn2d
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        bias_act.bias_act(torch.randn(2, 4), torch.randn(4), act='lrelu')
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        upfirdn2d.upfirdn2d(torch.randn(1, 1, 4, 4), upfirdn2d.setup_filter([1, 3, 3, 1]))
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        linear.linear(torch.randn(2, 4), torch.randn(3, 4))
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        conv.conv2d_nhwc(torch.randn(1, 4, 4, 4), torch.randn(4, 4, 1, 1))
    with pytest.raises(RuntimeError):
        bias_act.bias_act(torch.randn(2, 4), None, impl='ref')


def test_argument_validation_without_gpu():
    """C-side checks fire before any launch, so they are testable on a CPU-only box (null pointers, bad shapes)."""
    import ctypes
    from layoutdetr_amd import _lib
    lib = _lib.load()
    rc = lib.ldetr_bias_act_f32(None, None, None, None, None, None, 16, 0, 1, 0, 3, 0.2, 1.0, -1.0, None)
    assert rc != 0 and b'non-null' in lib.ldetr_last_error()
    rc = lib.ldetr_attention_fwd_f32(None, 0, None, 0, None, 0, None, None, 0, None, 1, 8, 9, 9, 48, 1.0, 0.0, 0, None, 0, None)
    assert rc != 0 and b'head_dim' in lib.ldetr_last_error()
    rc = lib.ldetr_lsap_f64(ctypes.c_void_p(8), 1, 99, 0, ctypes.c_void_p(8), ctypes.c_void_p(8), None)
    assert rc != 0 and b'n must be' in lib.ldetr_last_error()
    assert lib.ldetr_bias_act_f32(None, None, None, None, None, None, 0, 0, 1, 0, 3, 0.2, 1.0, -1.0, None) == 0  # empty input


def test_layout_losses_and_position_encoding_match_reference_golden():
    from layoutdetr_amd.detr_util.misc import NestedTensor
    from layoutdetr_amd.metrics import metric_layoutnet as M
    from layoutdetr_amd.training.detr_position_encoding import PositionEmbeddingSine
    d = load('losses')

    def close(a, b, tol=5e-5):
        assert ((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-12)).item() <= tol
    for nm, fn in [('overlap', M.compute_overlap), ('alignment', M.compute_alignment)]:
        b = d['bbox'].clone().requires_grad_(True)
        v = fn(b, d['mask']); v.sum().backward()
        close(v, d[nm]); close(b.grad, d['d_' + nm])
    b = d['bbox'].clone().requires_grad_(True)
    v = M.generalized_iou_loss(b[d['mask']], d['real'][d['mask']]); v.backward()
    close(v, d['giou']); close(b.grad, d['d_giou'])
    p = load('pos_encoding')
    pe = PositionEmbeddingSine(128, normalize=True)
    close(pe(NestedTensor(torch.zeros(2, 1, 4, 5), p['mask'])), p['pos'], 1e-6)


def test_module_surface_matches_reference_contract():
    """Seam 1: constructor kwargs of train.py:250-261 + training_loop.py:127-132, attributes callers touch, UP-DETR key names."""
    from layoutdetr_amd.training.networks_detr import Discriminator, Generator, TextFeatures, split_list
    from layoutdetr_amd.training import training_loop
    kw = dict(num_bbox_labels=8, img_channels=3, img_height=64, img_width=64, c_dim=0, background_size=64,
              f_dim=256, num_heads=4, num_layers=8, bert_f_dim=768, bert_num_heads=4, bert_num_encoder_layers=12,
              bert_num_decoder_layers=2, im_f_dim=512)
    G = Generator(z_dim=4, **kw)
    D = Discriminator(**kw)
    assert G.z_dim == 4 and hasattr(G, 'text_encoder') and hasattr(D, 'text_encoder')
    G.text_encoder.requires_grad_(False)
    keys = set(G.state_dict())
    for k in ['backbone.0.body.conv1.weight', 'backbone.0.body.layer4.2.bn3.running_var', 'input_proj.weight',
              'transformer.encoder.layers.5.self_attn.in_proj_weight', 'transformer.decoder.layers.0.multihead_attn.out_proj.bias',
              'transformer.decoder.norm.weight', 'bbox_embed.layers.2.weight', 'fc_in.layers.0.weight']:
        assert k in keys, k
    dk = set(D.state_dict())
    for k in ['enc_transformer.token', 'dec_transformer.layers.5.linear2.weight', 'enc_transformer_uncond.core.layers.0.norm1.weight',
              'bg_decoder.synthesis.b64.conv0.affine.weight', 'bg_decoder.mapping.fc7.bias', 'bg_decoder.synthesis.b4.const', 'pos_token_uncond']:
        assert k in dk, k
    assert G.state_dict()['backbone.0.body.layer2.0.conv2.weight'].shape == (128, 128, 3, 3)
    assert split_list(list(range(5)), 2) == [[0, 1], [2, 3], [4]] and training_loop.split_list is split_list
    tf = TextFeatures(torch.zeros(4, 9, 768), torch.zeros(4, 9, dtype=torch.int64))
    assert len(tf[:2]) == 2


def test_flat_module_views_and_grad_accumulation():
    from layoutdetr_amd.training.training_loop import FlatModule
    torch.manual_seed(0)
    m = torch.nn.Sequential(torch.nn.Linear(5, 3), torch.nn.Conv2d(4, 6, 3))
    m[1].weight.data = m[1].weight.data.to(memory_format=torch.channels_last)
    ref = [p.detach().clone() for p in m.parameters()]
    fm = FlatModule(m)
    for p, r in zip(m.parameters(), ref):
        assert torch.equal(p, r) and p.stride() == r.stride()
        assert fm.flat.data_ptr() <= p.data_ptr() < fm.flat.data_ptr() + 4 * fm.total
    x = torch.randn(2, 5); img = torch.randn(2, 4, 5, 5)
    (m[0](x).sum() + m[1](img).sum()).backward()
    g1 = fm.gflat.clone()
    assert g1.abs().sum() > 0
    (m[0](x).sum() + m[1](img).sum()).backward()     # accumulates in place into the flat buffer
    assert torch.allclose(fm.gflat, 2 * g1)
    fm.zero_grad()
    assert fm.gflat.abs().sum() == 0 and all(p.grad.abs().sum() == 0 for p in m.parameters())
    assert all(off % 4 == 0 for off in fm.offsets)   # 16-byte aligned segments


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _dp_worker(rank, world, port, q):
    import torch.distributed as dist
    from layoutdetr_amd.training.training_loop import DataParallelStep, FlatModule
    from oracle import losses_ref
    dist.init_process_group('gloo', init_method=f'tcp://127.0.0.1:{port}', rank=rank, world_size=world)
    torch.manual_seed(0)
    m = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3))
    fm = FlatModule(m)
    g = torch.Generator().manual_seed(100 + rank)
    local = torch.randn(fm.total, generator=g)
    if rank == 0:
        local[3] = float('nan'); local[5] = float('inf')
    if rank == 1:
        local[7] = -float('inf')
    fm.gflat.copy_(local)
    dp = DataParallelStep(world_size=world, bucket_bytes=64)   # 16-float buckets: exercises the bucket loop
    # segment by segment, as the staged backward hands the flat gradient over (training_loop.staged_backward): last segment first
    cuts = [0, 11, 30, fm.total]
    for lo, hi in reversed(list(zip(cuts[:-1], cuts[1:]))):
        dp.exchange_async(fm.gflat, lo, hi)
    dp.finish()
    # expected: sum over ranks; the /world + nan_to_num that the fused Adam kernel applies is checked through the oracle
    parts = [torch.randn(fm.total, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)]
    parts[0][3] = float('nan'); parts[0][5] = float('inf'); parts[1][7] = -float('inf')
    expect_sum = sum(parts)
    same = torch.equal(torch.nan_to_num(fm.gflat, nan=7.0), torch.nan_to_num(expect_sum, nan=7.0))
    post = losses_ref.dp_postprocess(fm.gflat, world)
    ok_post = (post[3] == 0) and (post[5] == 1e5) and (post[7] == -1e5) and torch.isfinite(post).all().item()
    # the stage count of the overlapped backward is derived from measured stage lengths: the ranks must derive it from the SAME numbers
    # (rank 0 measures a 1.6 ms shortest stage -> 3 stages, rank 1 1.4 ms -> 2: different segment cuts = mismatched collectives)
    from layoutdetr_amd.training.training_loop import backward_stage_count
    agreed = dp.agree_min([5.0, 1.6 if rank == 0 else 1.4, 2.0 + rank])
    ok_agree = agreed == [5.0, 1.4, 2.0] and backward_stage_count(16, agreed) == 2 and dp.agree_min(None) is None
    q.put((rank, bool(same), bool(ok_post) and ok_agree, fm.gflat.nan_to_num(7.0).sum().item()))
    dist.destroy_process_group()


class _TrunkLike(torch.nn.Module):
    """Parameter names / order of the generator as FlatModule.stage_segments reads them: direct parameters, a ResNet-like `backbone.0.body` with
    layer1..layer4, heads behind it."""

    def __init__(self):
        super().__init__()
        self.token = torch.nn.Parameter(torch.zeros(5))
        body = torch.nn.Module()
        body.conv1 = torch.nn.Linear(3, 4)
        for i, n in enumerate((6, 10, 14, 9)):
            setattr(body, f'layer{i + 1}', torch.nn.Sequential(torch.nn.Linear(n, n + 1), torch.nn.Linear(n + 1, 3)))
        bb = torch.nn.Module(); bb.body = body
        self.backbone = torch.nn.Sequential(bb)
        self.head = torch.nn.Linear(11, 7)
        self.text_encoder = torch.nn.Linear(2, 2)      # frozen: outside the flat buffers


def _sequence_worker(rank, world, port, q):
    """Every rank 'measures' different stage lengths (some on either side of the 1.5 ms threshold), agrees on them, derives the stage count and
    exchanges its flat gradient stage by stage; the sequence of all_reduce calls (offset, length) it ISSUED is returned."""
    import torch.distributed as dist
    from layoutdetr_amd.training import training_loop as tl
    dist.init_process_group('gloo', init_method=f'tcp://127.0.0.1:{port}', rank=rank, world_size=world)
    torch.manual_seed(0)
    fm = tl.FlatModule(_TrunkLike())
    dp = tl.DataParallelStep(world_size=world, bucket_bytes=4 * 40)        # 40-float buckets: several per stage
    issued = []
    real = dist.all_reduce

    def recording(t, *a, **k):
        issued.append((t.storage_offset(), t.numel()))
        return real(t, *a, **k)
    dist.all_reduce = recording
    out = []
    try:
        for case, mine in enumerate(([5.0, 1.45 + 0.02 * rank, 3.0], [5.0, 1.6 + 0.1 * rank, 1.7], [4.0 - 0.3 * rank, 9.0, 2.0 + rank])):
            agreed = dp.agree_min(mine)
            issued.clear()                       # (the agreement itself is a collective too: same on every rank by construction)
            n = tl.backward_stage_count(16, agreed)
            fm.gflat.copy_(torch.arange(fm.total, dtype=torch.float32) * (rank + 1))
            plan = dp.collective_plan(fm, n)
            for stage in fm.stage_segments(n):       # what staged_backward does after each stage of the backward
                for lo, hi in stage:
                    dp.exchange_async(fm.gflat, lo, hi)
            dp.finish()
            ok_sum = torch.equal(fm.gflat, torch.arange(fm.total, dtype=torch.float32) * (world * (world + 1) / 2))
            out.append((case, n, list(issued), [(lo, hi - lo) for lo, hi in plan], bool(ok_sum)))
    finally:
        dist.all_reduce = real
    q.put((rank, out))
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 4, 8])
def test_collective_sequence_is_identical_on_every_rank(world):
    """RCCL readiness without hardware: segment bounds, bucket order and stage count of the overlapped exchange on 2 / 4 / 8 ranks (gloo) whose
    LOCAL stage measurements differ -- every rank must issue the same all_reduce sequence, equal to DataParallelStep.collective_plan, covering the
    flat buffer exactly once, for the 2-stage and the 3-stage cut."""
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0)); port = sk.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    ps = [ctx.Process(target=_sequence_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(world))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    stage_counts = set()
    for case in range(3):
        ref = res[0][case]
        stage_counts.add(ref[1])
        for r in range(world):
            c, n, issued, plan, ok_sum = res[r][case]
            assert (c, n) == (case, ref[1]), f'rank {r} chose {n} stages, rank 0 {ref[1]}'
            assert issued == ref[2], f'rank {r} issued a different collective sequence in case {case}'
            assert issued == plan, 'the issued sequence is not the planned one'
            assert ok_sum, 'SUM over the ranks is wrong'
        covered = sorted(ref[2])
        assert covered[0][0] == 0 and all(a[0] + a[1] == b[0] for a, b in zip(covered, covered[1:])), 'the segments do not tile the flat buffer'
    assert stage_counts == {2, 3}, f'the cases must exercise both cuts, got {stage_counts}'


def test_regulariser_phase_shares_the_optimiser_state_and_partitions_the_flat_buffer():
    """training_loop.Phase(share=main): the lazy regulariser phase of training_loop.py:190-197 uses the main phase's optimiser.  torch.optim.Adam skips
    parameters whose .grad is None and counts steps per parameter, so after the first regulariser step the flat buffer has two groups of ranges with
    different step counts: `touched_runs` finds the ranges (any non-zero gradient element), `step_ranges` partitions the buffer for either kind of step."""
    from layoutdetr_amd.training import training_loop as tl
    torch.manual_seed(0)
    m = _TrunkLike()
    main = tl.Phase('Dmain', m, lr=1e-3, betas=(0.0, 0.99), reg_interval=16)
    reg = tl.Phase('Dreg', m, share=main, interval=16)
    assert reg.fm is main.fm and reg.m is main.m and reg.v is main.v and reg.main is main and reg.interval == 16
    assert abs(main.lr - 1e-3 * 16 / 17) < 1e-12 and reg.lr == main.lr and abs(main.betas[1] - 0.99 ** (16 / 17)) < 1e-12
    fm = main.fm
    assert main.step_ranges(False) == [(0, fm.total, 0)]
    # a regulariser that reaches the token, layer2 and the head only
    fm.zero_grad()
    hit = [i for i, n in enumerate(fm.names) if n == 'token' or 'layer2' in n or n.startswith('head')]
    for i in hit:
        fm.gflat[fm.offsets[i]:fm.offsets[i] + fm.params[i].numel()] = 1.0
    runs = main.touched_runs()
    covered = torch.zeros(fm.total, dtype=torch.bool)
    for lo, hi in runs:
        assert 0 <= lo < hi <= fm.total
        covered[lo:hi] = True
    for i, (p, o) in enumerate(zip(fm.params, fm.offsets)):
        assert bool(covered[o:o + p.numel()].all()) == (i in hit), fm.names[i]
    assert runs == sorted(runs) and all(a[1] < b[0] for a, b in zip(runs, runs[1:])), 'adjacent parameters merge into one range'
    main.reg_runs, main.step, main.reg_steps = runs, 5, 2
    r = reg.step_ranges(True)
    assert [(lo, hi) for lo, hi, _ in r] == runs and all(st == 7 for _, _, st in r)
    mm = main.step_ranges(False)
    assert mm[0][0] == 0 and mm[-1][1] == fm.total and all(a[1] == b[0] for a, b in zip(mm, mm[1:])), 'the main step covers the buffer exactly once'
    assert {st for _, _, st in mm} == {5, 7} and [(lo, hi) for lo, hi, st in mm if st == 7] == runs


def test_bench_preflight_refuses_more_local_ranks_than_gpus():
    """`bench.py --gpus N` under torch.distributed.run on a node that shows fewer GPUs than local ranks: ONE parsable line on rank 0, exit code 2,
    before any rendezvous (here: no GPU at all)."""
    import json
    import subprocess
    env = dict(os.environ, RANK='0', LOCAL_RANK='0', WORLD_SIZE='2', LOCAL_WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT='29999',
               HIP_VISIBLE_DEVICES='', CUDA_VISIBLE_DEVICES='')
    env.pop('LDETR_BENCH_SHARE_GPU', None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 2, (r.returncode, r.stderr[-500:])
    line = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(line) == 1
    d = json.loads(line[0])
    assert d['value'] is None and d['n_gpus'] == 2 and 'GPU(s) visible' in d['error']


def test_dp_gradient_exchange_gloo_world2():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] and r[2] for r in res), res
    assert res[0][3] == res[1][3]          # both ranks hold bit-identical reduced gradients


def test_bench_batch_helpers():
    import bench
    bt = bench.make_batch(3, 32, 'cpu', 5)
    assert bt['bbox_real'].shape == (3, 9, 4) and bt['background'].shape == (3, 3, 32, 32)
    assert (bt['bbox_real'][..., :2] >= 0.2).all() and (bt['bbox_real'][..., 2:] <= 0.4).all()
    assert not bt['padding_mask'].any()


def test_register_budgeted_kernels_use_no_scratch():
    """The contraction-engine / attention kernels are designed around their register budget; the build records clang's
    per-kernel resource usage and refuses scratch in them (guards against silent 10x regressions from an innocent edit)."""
    from layoutdetr_amd import build
    build.build(verbose=False)
    import glob
    files = glob.glob(os.path.join(build.OBJDIR, '*.resources.txt'))
    assert files, 'no resource reports next to the objects'
    n = 0
    for f in files:
        for line in open(f):
            name = line.split()[0]
            if any(k in name for k in build.NO_SCRATCH):
                n += 1
                assert ' scratch=0 ' in line, line
    assert n > 50


def test_resample_coefficients_match_pillow_restatement():
    """ldetr_resample_coeffs (host-side, part of the C-ABI) reproduces the oracle's Lanczos windows bit-for-bit: bounds and the 22-bit
    fixed-point weights, for down-scaling, up-scaling, equal sizes and extreme ratios; and it rejects a short weight buffer."""
    import ctypes
    import numpy as np
    from layoutdetr_amd import _lib
    from oracle import resample_ref
    lib = _lib.load()
    for i, o in [(1024, 256), (700, 256), (37, 256), (48, 48), (129, 31), (64, 512), (1000, 7), (1, 5)]:
        ks = ctypes.c_int(0)
        assert lib.ldetr_resample_coeffs(i, o, None, None, 0, ctypes.byref(ks)) == 0
        b = np.zeros((o, 2), np.int32); k = np.zeros((ks.value, o), np.int32)
        assert lib.ldetr_resample_coeffs(i, o, b.ctypes.data_as(ctypes.c_void_p), k.ctypes.data_as(ctypes.c_void_p), k.size, ctypes.byref(ks)) == 0
        rb, rk, rks = resample_ref.precompute_coeffs(i, o)
        assert ks.value == rks and np.array_equal(b, rb) and np.array_equal(k.T, rk), (i, o)
        assert int(k.sum(0).min()) > (1 << 22) - 64 and int(k.sum(0).max()) < (1 << 22) + 64   # every window sums to ~1.0
    assert lib.ldetr_resample_coeffs(1024, 256, b.ctypes.data_as(ctypes.c_void_p), k.ctypes.data_as(ctypes.c_void_p), 10, ctypes.byref(ks)) != 0
    assert b'too small' in lib.ldetr_last_error()


def test_wordpiece_tokenizer_matches_transformers(tmp_path):
    """training/tokenizer.py == transformers.BertTokenizer (+ the two tokens blip.init_tokenizer adds) id for id: casing, accents,
    punctuation splitting, CJK, unknown words, truncation, padding; bos / pad ids as networks_detr.py:172-173 uses them."""
    from transformers import BertTokenizer
    from layoutdetr_amd.training.tokenizer import BertWordPieceTokenizer, texts_to_tokens
    vocab = ['[PAD]', '[unused0]', '[UNK]', '[CLS]', '[SEP]', '[MASK]', 'sale', 'up', 'to', '50', '%', 'off', 'shop', 'now', 'new', 'arrival', '##s', 'this', 'week',
             'free', 'ship', '##ping', 'on', 'order', 'over', '$', '25', '!', 'sign', 'limit', '##ed', 'time', 'only', 'x', 'ok', 'cafe', '##teria', 'a', 'b', '##c',
             ',', '.', '-', '中', '国']
    vf = tmp_path / 'vocab.txt'
    vf.write_text('\n'.join(vocab) + '\n', encoding='utf-8')
    mine = BertWordPieceTokenizer(str(vf))
    ref = BertTokenizer(vocab={t: i for i, t in enumerate(vocab)})
    ref.add_special_tokens({'bos_token': '[DEC]'}); ref.add_special_tokens({'additional_special_tokens': ['[ENC]']})
    texts = ['Sale', 'Up to 50% off', 'Shop now', 'New arrivals this week', 'x', 'Free shipping on orders over $25', 'Sign up', 'Limited time only!', 'ok',
             'Café-teria, abc.  a\tb', '中国 sale', 'unknownword zzz', '', 'Free shipping on orders over $25 new arrivals this week sale sale sale']
    for L in (16, 8):
        r = ref(texts, padding='max_length', truncation=True, max_length=L, return_tensors='pt')
        ids, am = mine(texts, max_length=L)
        assert torch.equal(ids, r.input_ids) and torch.equal(am, r.attention_mask)
    assert mine.bos_token_id == ref.bos_token_id == len(vocab) and mine.pad_token_id == ref.pad_token_id == 0 and len(mine) == len(ref)
    tok = texts_to_tokens(mine, [texts[:3], texts[3:6]], 256)
    assert tok.input_ids.shape[:2] == (2, 3) and tok.input_ids.shape[2] == int(tok.attention_mask.sum(-1).max())
    assert tok.text_len.tolist() == [[4, 13, 8], [22, 1, 32]]


def test_dropin_aliases_and_training_loop_signature():
    """`training.networks_detr.Generator` etc. resolve to this package after dropin.install(); training_loop takes exactly the
    reference's keyword arguments (training/training_loop.py:63-99), so train.py's `training_loop.training_loop(rank=rank, **c)` binds."""
    import importlib
    import inspect
    from layoutdetr_amd import dropin
    dropin.install()
    try:
        _dropin_checks(importlib, inspect)
    finally:
        dropin.uninstall()          # (install() also switches the constructors to the reference's defaults, process-wide)
    assert 'training.networks_detr' not in sys.modules
    from layoutdetr_amd.training import networks_detr
    assert networks_detr.REFERENCE_DEFAULTS is False


def _dropin_checks(importlib, inspect):
    nd = importlib.import_module('training.networks_detr')
    assert nd.Generator.__module__ == 'layoutdetr_amd.training.networks_detr' and hasattr(nd, 'split_list')
    assert importlib.import_module('training.loss').StyleGAN2Loss.__module__ == 'layoutdetr_amd.training.loss'
    for m in ('bias_act', 'upfirdn2d', 'conv2d_resample', 'conv2d_gradfix'):
        assert importlib.import_module('torch_utils.ops.' + m).__name__ == 'layoutdetr_amd.torch_utils.ops.' + m
    ref_args = ['run_dir', 'training_set_kwargs', 'validation_set_kwargs', 'data_loader_kwargs', 'G_kwargs', 'D_kwargs', 'G_opt_kwargs', 'D_opt_kwargs',
                'augment_kwargs', 'loss_kwargs', 'metrics', 'random_seed', 'num_gpus', 'rank', 'batch_size', 'batch_gpu', 'ema_kimg', 'ema_rampup',
                'G_reg_interval', 'D_reg_interval', 'augment_p', 'ada_target', 'ada_interval', 'ada_kimg', 'total_kimg', 'kimg_per_tick',
                'image_snapshot_ticks', 'network_snapshot_ticks', 'resume_pkl', 'resume_kimg', 'cudnn_benchmark', 'abort_fn', 'progress_fn']
    tl = importlib.import_module('training.training_loop')
    assert list(inspect.signature(tl.training_loop).parameters) == ref_args
    # the Generator / Discriminator constructors accept every keyword train.py passes (train.py:250-261 + training_loop.py:127-132)
    kw = dict(z_dim=4, f_dim=256, num_heads=4, num_layers=8, bert_f_dim=768, bert_num_heads=4, bert_num_encoder_layers=12, bert_num_decoder_layers=2, im_f_dim=512,
              num_bbox_labels=8, img_channels=3, img_height=64, img_width=64, background_size=64, c_dim=0)
    inspect.signature(nd.Generator.__init__).bind(None, **kw)
    kw.pop('z_dim'); inspect.signature(nd.Discriminator.__init__).bind(None, **kw)
    sampler = tl.InfiniteSampler(list(range(10)), rank=1, num_replicas=2, shuffle=False)
    it = iter(sampler); assert [next(it) for _ in range(6)] == [1, 3, 5, 7, 9, 1]
    # under the reference's driver an unspecified text_mode is what the reference always builds; without the vocabulary that fails HERE
    assert nd.REFERENCE_DEFAULTS is True
    saved = os.environ.pop('LDETR_BERT_VOCAB', None)
    try:
        with pytest.raises(RuntimeError, match='LDETR_BERT_VOCAB'):
            nd._resolve_text_mode(None, nd._build_tokenizer(None))
        assert nd._resolve_text_mode('features', None) == 'features'
    finally:
        if saved is not None:
            os.environ['LDETR_BERT_VOCAB'] = saved


def _stats_worker(rank, world, port, q):
    import torch.distributed as dist
    from layoutdetr_amd.training import training_loop as tl
    dist.init_process_group('gloo', init_method=f'tcp://127.0.0.1:{port}', rank=rank, world_size=world)
    try:
        # (1) per-tick statistics: float64 [count, sum, sum of squares] per name, ONE all-reduce of the stacked matrix (training_stats.py:232-254)
        c = tl.StatsCollector(device=torch.device('cpu'), world=world)
        vals = {0: [1.0, 2.0, 3.0], 1: [10.0]}[rank]
        c.report('Loss/scores/fake', torch.tensor(vals)); c.report('Loss/scores/fake', torch.tensor([4.0 + rank]))
        c.report('Loss/G/loss', torch.tensor(2.0 * (rank + 1))); c.report('Loss/empty', [])
        d = c.update()
        allv = np.array([1.0, 2.0, 3.0, 4.0, 10.0, 5.0])
        ok_stats = (d['Loss/scores/fake']['num'] == 6 and abs(d['Loss/scores/fake']['mean'] - allv.mean()) < 1e-12
                    and abs(d['Loss/scores/fake']['std'] - allv.std()) < 1e-9 and abs(d['Loss/G/loss']['mean'] - 3.0) < 1e-12 and d['Loss/empty']['num'] == 0)
        c.report('Loss/scores/fake', torch.tensor([7.0]))
        d2 = c.update()      # the next interval holds only what was reported since
        ok_stats = ok_stats and d2['Loss/scores/fake']['num'] == 2 and d2['Loss/scores/fake']['mean'] == 7.0 and d2['Loss/G/loss']['num'] == 0
        # (2) one flat broadcast per dtype group (training_loop.py:176-179), incl. bool / int64 buffers and a channels_last parameter
        torch.manual_seed(rank)
        m = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3), torch.nn.BatchNorm2d(4), torch.nn.Linear(5, 2))
        m[0].weight.data = m[0].weight.data.contiguous(memory_format=torch.channels_last)
        m.register_buffer('flag', torch.tensor([rank == 0, True]))
        before = [t.detach().clone() for t in m.state_dict().values()]
        consistent_before = True
        try:
            tl.check_ddp_consistency(m)
        except AssertionError as e:
            consistent_before = False
            named = str(e)
        tl.broadcast_module(m, src=0)
        torch.manual_seed(0)
        ref = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3), torch.nn.BatchNorm2d(4), torch.nn.Linear(5, 2))
        want = dict(ref.state_dict(), flag=torch.tensor([True, True]))
        same = all(torch.equal(v, want[k]) for k, v in m.state_dict().items())
        layout_kept = m[0].weight.is_contiguous(memory_format=torch.channels_last)
        tl.check_ddp_consistency(m)          # passes on every rank now
        # (3) a divergence is reported with the reference's naming (ClassName.tensor name), on the rank that differs; ignore_regex skips it
        if rank == 1:
            m[2].bias.data[0] += 1.0
        caught = ''
        try:
            tl.check_ddp_consistency(m)
        except AssertionError as e:
            caught = str(e)
        tl.check_ddp_consistency(m, ignore_regex=r'.*\.2\.bias')
        q.put((rank, bool(ok_stats), bool(same and layout_kept), consistent_before if rank == 0 else (not consistent_before and 'Sequential.' in named),
               caught == ('' if rank == 0 else 'Sequential.2.bias')))
    finally:
        dist.destroy_process_group()


def test_stats_allreduce_flat_broadcast_and_consistency_check_gloo_world2():
    """SURVEY 8e "also needed": the float64 statistics all-reduce (training_stats.py:232-254), the initial broadcast (one collective per
    dtype group instead of one per tensor) and check_ddp_consistency (misc.py:183-194), two ranks over gloo."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_stats_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(all(r[1:]) for r in res), res


def test_p3_entry_points_validate_before_launching():
    """The plane-format convolution entry points (csrc/p3_engine.hip) check geometry and pointers on the host: testable without a GPU."""
    import ctypes
    from layoutdetr_amd import _lib
    lib = _lib.load()
    P = ctypes.c_void_p
    rc = lib.ldetr_p3_conv2d_fwd(None, 1, 8, 8, 32, None, 64, 1, 1, 1, 0, None, None, None, None)
    assert rc != 0 and b'null operand' in lib.ldetr_last_error()
    rc = lib.ldetr_p3_conv2d_fwd(P(64), 1, 8, 8, 24, P(64), 64, 1, 1, 1, 0, None, P(64), None, None)          # Cin not a multiple of 32
    assert rc != 0 and b'unsupported geometry' in lib.ldetr_last_error()
    rc = lib.ldetr_p3_conv2d_fwd(P(64), 1, 8, 8, 32, P(64), 64, 3, 3, 3, 1, None, P(64), None, None)          # stride 3
    assert rc != 0 and b'unsupported geometry' in lib.ldetr_last_error()
    rc = lib.ldetr_p3_conv2d_fwd(P(64), 4096, 512, 512, 256, P(64), 64, 1, 1, 1, 0, None, P(64), None, None)  # beyond 31-bit buffer offsets
    assert rc != 0 and b'too large' in lib.ldetr_last_error()
    rc = lib.ldetr_p3_conv2d_fwd_dual(P(64), None, 1, 8, 8, 32, P(64), P(64), 64, 1, 1, 1, 0, None, None, P(64), None, P(64), None, None)
    assert rc != 0 and b'second set' in lib.ldetr_last_error()
    rc = lib.ldetr_p3_conv2d_bwd_weight(P(64), 1, 8, 8, 32, P(64), 48, 1, 1, 1, 0, None, P(64), None)         # Cout not a multiple of 32
    assert rc != 0 and b'unsupported geometry' in lib.ldetr_last_error()
    rc = lib.ldetr_p3_conv2d_bwd_pair(P(64), 1, 8, 8, 64, P(64), P(64), 32, 1, 1, 1, 0, 8, 8, None, None, None, None, P(64), None, None)
    assert rc != 0 and b'null operand' in lib.ldetr_last_error()                                              # neither dx_p3 nor dx_f32
    rc = lib.ldetr_p3_split_f32(P(64), 12, P(64), 4, 12, None)                                               # C not a multiple of 8
    assert rc != 0 and b'multiple of 8' in lib.ldetr_last_error()
    assert lib.ldetr_p3_split_f32(P(64), 16, P(64), 0, 16, None) == 0                                         # empty input


def test_backward_stage_count_and_two_stage_segments(monkeypatch):
    """training_loop.backward_stage_count (trunk | rest at <= 4 samples per GPU, three stages above, LDETR_BACKWARD_STAGES overrides) and
    FlatModule.stage_segments(2): the two-stage segments tile the flat buffer and merge the three-stage trunk segments."""
    from layoutdetr_amd.training import training_loop as tl
    monkeypatch.delenv('LDETR_BACKWARD_STAGES', raising=False)
    assert [tl.backward_stage_count(b) for b in (1, 2, 4, 5, 16)] == [2, 2, 2, 3, 3]
    # measured stage lengths [rest, layer3-4, layer1-2] decide when given: three stages only if each covers the host's issue latency
    assert tl.backward_stage_count(16, [6.0, 3.1, 2.2]) == 3 and tl.backward_stage_count(16, [6.0, 3.1, 0.9]) == 2 and tl.backward_stage_count(2, [2.0, 1.6, 1.5]) == 3
    monkeypatch.setenv('LDETR_BACKWARD_STAGES', '3')
    assert tl.backward_stage_count(2) == 3
    monkeypatch.setenv('LDETR_BACKWARD_STAGES', '2')
    assert tl.backward_stage_count(16) == 2
    monkeypatch.setenv('LDETR_BACKWARD_STAGES', 'junk')
    assert tl.backward_stage_count(16) == 3

    class Body(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.conv1 = torch.nn.Linear(3, 4)
            for li in range(1, 5):
                setattr(self, f'layer{li}', torch.nn.Linear(4, 4))

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.pos_token = torch.nn.Parameter(torch.zeros(5))
            self.backbone = torch.nn.Sequential(torch.nn.Sequential())
            self.backbone[0].body = Body()
            self.head = torch.nn.Linear(4, 2)
    fm = tl.FlatModule(Net())
    s3, s2 = fm.stage_segments(3), fm.stage_segments(2)
    assert len(s3) == 3 and len(s2) == 2 and s2[0] == s3[0]
    (lo3, hi3), = s3[2]; (lo2, hi2), = s3[1]; (lo, hi), = s2[1]
    assert (lo, hi) == (lo3, hi2) and hi3 == lo2, 'the two-stage trunk segment is layer1-2 followed by layer3-4'
    ranges = sorted(r for st in s2 for r in st)
    assert ranges[0][0] == 0 and ranges[-1][1] == fm.total and all(a[1] == b[0] for a, b in zip(ranges[:-1], ranges[1:]))
