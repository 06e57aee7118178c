import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
dev = torch.device('cuda:0')
which = sys.argv[1]
def run(fn):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2): fn()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): fn()
    torch.cuda.synchronize()
    for _ in range(2): g.replay()
    torch.cuda.synchronize(); print(which, 'ok', flush=True)
if which == 'emb':
    e = torch.nn.Embedding(30524, 768, padding_idx=0).to(dev); ids = torch.randint(1, 30000, (144, 40), device=dev)
    def fn():
        e.weight.grad = None
        e(ids).square().sum().backward()
    run(fn)
if which == 'ce':
    x = torch.randn(5616, 30524, device=dev, requires_grad=True); t = torch.randint(0, 30524, (5616,), device=dev); t[::3] = -100
    def fn():
        x.grad = None
        F.cross_entropy(x, t, ignore_index=-100, label_smoothing=0.1).backward()
    run(fn)
if which == 'lin':
    from layoutdetr_amd.hip.linear import linear
    w = torch.randn(30524, 768, device=dev, requires_grad=True); b = torch.zeros(30524, device=dev, requires_grad=True)
    x = torch.randn(5616, 768, device=dev, requires_grad=True)
    def fn():
        w.grad = b.grad = x.grad = None
        linear(x, w, b).square().mean().backward()
    run(fn)
if which in ('lm', 'lm_nodrop', 'lm_fwd'):
    from layoutdetr_amd.training import med
    cfg = med.BertConfig(num_hidden_layers=2, num_attention_heads=4, vocab_size=30524)
    m = med.BertLMHeadModel(cfg).to(dev)
    m.train() if which != 'lm_nodrop' else m.eval()
    S, T = 144, 40
    ids = torch.randint(1000, 30000, (S, T), device=dev); am = torch.ones(S, T, dtype=torch.long, device=dev); am[:, T // 2:] = 0; ids = ids * am
    ids[:, 0] = 30522; labels = ids.masked_fill(ids == 0, -100)
    def fn():
        for p in m.parameters(): p.grad = None
        lo = m(ids, attention_mask=am, labels=labels).loss
        if which != 'lm_fwd': lo.backward()
    run(fn)
if which == 'gelu':
    from layoutdetr_amd.training.med import _GeluFn
    h = torch.randn(5616, 768, device=dev, requires_grad=True); b = torch.randn(768, device=dev, requires_grad=True)
    def fn():
        h.grad = b.grad = None
        _GeluFn.apply(h, b).square().mean().backward()
    run(fn)
if which == 'attn':
    from layoutdetr_amd.hip.attention import _AttnPackedFn
    qkv = torch.randn(144 * 40, 3 * 768, device=dev, requires_grad=True); kpm = torch.zeros(144, 40, dtype=torch.uint8, device=dev)
    def fn():
        qkv.grad = None
        _AttnPackedFn.apply(qkv, None, kpm, 144, 4, 40, 0.1, True).square().mean().backward()
    run(fn)
