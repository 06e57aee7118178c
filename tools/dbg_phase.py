import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import make_batch, to_device_batch
from layoutdetr_amd.training import training_loop as tl
from layoutdetr_amd.training.loss import StyleGAN2Loss
from layoutdetr_amd.training.networks_detr import Discriminator, Generator
dev = torch.device('cuda:0'); B = int(sys.argv[2]); bg = 256; which = sys.argv[1]; T = int(sys.argv[3])
torch.manual_seed(0)
kw = dict(num_bbox_labels=8, img_channels=3, img_height=bg, img_width=bg, c_dim=0, background_size=bg, bert_f_dim=768, im_f_dim=512,
          bert_num_heads=4, bert_num_encoder_layers=2, bert_num_decoder_layers=2, text_mode=os.environ.get('LDETR_TM', 'encoder+lm'))
import torch.nn.functional as F
from layoutdetr_amd.training import med
mode = os.environ.get('LDETR_DBG', '')
if mode:
    _orig = med.BertLMHeadModel.forward
    def patched(self, input_ids, attention_mask=None, labels=None, **kw):
        from types import SimpleNamespace
        from layoutdetr_amd.hip.linear import linear
        from layoutdetr_amd.hip.layernorm import add_layernorm
        cfg = self.config; B_, T_ = input_ids.shape; emb = self.bert.embeddings
        x = emb.word_embeddings(input_ids) + emb.position_embeddings(emb.position_ids[:, :T_])
        x2 = add_layernorm(x.reshape(-1, cfg.hidden_size), None, emb.LayerNorm.weight, emb.LayerNorm.bias, emb.LayerNorm.eps)
        kpm = (attention_mask == 0).to(torch.uint8).contiguous()
        if mode != 'nolayers':
            for layer in self.bert.encoder.layer:
                x2 = med._layer_train(layer, x2, B_, T_, kpm, True)
        hsh = x2.reshape(B_, T_, -1)[:, :-1].reshape(-1, cfg.hidden_size)
        if mode == 'nohead':
            return SimpleNamespace(loss=hsh.square().mean())
        logits = linear(hsh, self.cls.predictions.decoder.weight, self.cls.predictions.bias)
        if mode == 'noce':
            return SimpleNamespace(loss=logits.square().mean())
        return SimpleNamespace(loss=F.cross_entropy(logits, labels[:, 1:].reshape(-1), ignore_index=-100, label_smoothing=0.1))
    med.BertLMHeadModel.forward = patched
G = Generator(z_dim=4, **kw).train().requires_grad_(False).to(dev); D = Discriminator(**kw).train().requires_grad_(False).to(dev)
G.static_shapes = D.static_shapes = True
LR = float(os.environ.get('LDETR_LR', '1e-5')); pG = tl.Phase('Gmain', G, lr=LR); pD = tl.Phase('Dmain', D, lr=LR)
loss = StyleGAN2Loss(dev, G, D); dp = tl.DataParallelStep(1)
batch = to_device_batch(make_batch(B, bg, dev, 1), dev, os.environ.get('LDETR_TM', 'encoder+lm'), T)
phases = {'G': [pG], 'D': [pD], 'GD': [pG, pD]}[which]
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        tl.training_iteration(loss, phases, dp, batch, B, [torch.randn(B, 9, 4, device=dev) for _ in phases])
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize(); print('eager ok', flush=True)
from layoutdetr_amd.hip import core as _c
_ws = _c._workspace.get(torch.cuda.current_device())
if _ws is not None and os.environ.get('LDETR_DBG_COUNTERS'):
    cnt = _ws[:262144].view(torch.int32)
    nz = (cnt != 0).nonzero().flatten()
    print('non-zero counters after eager iterations:', nz.numel(), nz[:10].tolist(), cnt[nz[:10]].tolist(), flush=True)
if os.environ.get('LDETR_EAGER_ONLY'):
    sys.exit(0)
gi = tl.GraphedIteration(loss, phases, dp, batch, B, 4, capture_stream=side)
torch.cuda.synchronize(); print('captured', flush=True)
if os.environ.get('LDETR_DBG_EMPTY'):
    import gc; gc.collect(); torch.cuda.empty_cache(); print('cache emptied', flush=True)
NREP = int(os.environ.get('LDETR_DBG_REPLAYS', '3'))
for i in range(NREP):
    gi.run()
    if NREP > 10 and i % 50 != 49 and i != NREP - 1:
        continue
    torch.cuda.synchronize()
    mods = [ph.module for ph in phases]
    bad = [n for m in mods for n, q in m.named_parameters() if not torch.isfinite(q).all()]
    gmax = max(float(q.grad.abs().max()) for m in mods for q in m.parameters() if q.grad is not None)
    worst = sorted(((float(q.grad.abs().max()), n) for m in mods for n, q in m.named_parameters() if q.grad is not None), reverse=True)[:3]
    print('replay', i, 'non-finite params:', len(bad), bad[:4], 'max|grad|', gmax, worst, flush=True)
