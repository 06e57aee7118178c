"""Two EAGER G+D iterations at the bench workload (no hipGraph, no roofline leg): the process rocprofv3 --pmc wraps to collect HBM
traffic counters for every kernel of the step (profiles/r02_pmc_traffic.*).
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out_f -- python tools/pmc_step.py
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out_w -- python tools/pmc_step.py      (separate passes)
then  python tools/pmc_step.py --aggregate out_f out_w 2 > profiles/pmc_traffic.json   (stamped with the kernel sources' digest; bench.py quotes it only for the same build)"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def family(n):
    n = n.replace('void ', '')
    if 'ldetr::' in n:
        return n.split('(')[0].replace('ldetr::', '')
    return 'aten/other'


def aggregate(dir_f, dir_w, iters):
    out = {}
    for d, key in ((dir_f, 'fetch'), (dir_w, 'write')):
        for path in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
            with open(path) as fh:
                for r in csv.DictReader(fh):
                    fam = family(r['Kernel_Name'])
                    e = out.setdefault(fam, dict(fetch=0.0, write=0.0, launches=0))
                    # FETCH_SIZE / WRITE_SIZE are reported in KiB; FETCH_SIZE x2 for wide coalesced reads on gfx950 (MI355X_MICROARCH.md, HBM section)
                    v = float(r['Counter_Value']) * 1024.0
                    e[key] += (2.0 * v if key == 'fetch' else v) / iters
                    if key == 'fetch':
                        e['launches'] += 1.0 / iters
    eng = [k for k in out if k.startswith(('gemm_', 'p3_nt_', 'p3_c3_', 'p3_tn_', 'p3_bwd_pair_', 'conv3x3_c32', 'wgrad_c32', 'stem_conv', 'mha_small_', 'mha_cross_', 'ffn_', 'wgrad_multi'))]   # every kernel behind hip.core.engine_call
    tot = dict(fetch=sum(out[k]['fetch'] for k in eng), write=sum(out[k]['write'] for k in eng), launches=sum(out[k]['launches'] for k in eng))
    from layoutdetr_amd import build as kbuild
    return dict(csrc_digest=kbuild.source_digest(), note='rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes over tools/pmc_step.py (2 eager iterations; default B=16, 256x256, text features in); bytes per ITERATION; '
                     'FETCH_SIZE doubled (gfx950 128-B requests tallied at 64 B), WRITE_SIZE as reported', engine_total=tot,
                engine_bytes_per_launch=(tot['fetch'] + tot['write']) / max(tot['launches'], 1), by_kernel={k: out[k] for k in sorted(out, key=lambda k: -(out[k]['fetch'] + out[k]['write']))})


def main():
    if len(sys.argv) > 1 and sys.argv[1] == '--aggregate':
        print(json.dumps(aggregate(sys.argv[2], sys.argv[3], float(sys.argv[4])), indent=1))
        return
    import torch
    import bench
    from layoutdetr_amd.training import training_loop as tl
    from layoutdetr_amd.training.loss import StyleGAN2Loss
    from layoutdetr_amd.training.networks_detr import Discriminator, Generator
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument('--b', type=int, default=16); ap.add_argument('--bg', type=int, default=256); ap.add_argument('--text-mode', default='features')
    a = ap.parse_args()
    b, bg = a.b, a.bg
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    kw = dict(num_bbox_labels=8, img_channels=3, img_height=bg, img_width=bg, c_dim=0, background_size=bg, bert_f_dim=768, im_f_dim=512,
              bert_num_heads=4, bert_num_encoder_layers=12, bert_num_decoder_layers=2, text_mode=a.text_mode)
    G = Generator(z_dim=4, **kw).train().requires_grad_(False).to(dev)
    D = Discriminator(**kw).train().requires_grad_(False).to(dev)
    G.static_shapes = D.static_shapes = True
    pG, pD = tl.Phase('Gmain', G, lr=1e-5), tl.Phase('Dmain', D, lr=1e-5)
    loss = StyleGAN2Loss(dev, G, D, share_D_trunk='iteration')     # the bench headline's setting
    dp = tl.DataParallelStep(1)
    batch = bench.to_device_batch(bench.make_batch(b, bg, dev, 1), dev, a.text_mode)
    for _ in range(2):
        z = [torch.randn(b, 9, 4, device=dev) for _ in range(2)]
        tl.training_iteration(loss, [pG, pD], dp, batch, b, z)
    torch.cuda.synchronize()


if __name__ == '__main__':
    main()
