import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from layoutdetr_amd.training import med
dev = torch.device('cuda:0')
S, T = int(sys.argv[1]), int(sys.argv[2])
cfg = med.BertConfig(num_hidden_layers=2, num_attention_heads=4, vocab_size=30524)
m = med.BertLMHeadModel(cfg).train().to(dev)
ids = torch.randint(1000, 30000, (S, T), device=dev); am = torch.ones(S, T, dtype=torch.long, device=dev); am[:, T // 2:] = 0; ids = ids * am
ids[:, 0] = 30522
labels = ids.masked_fill(ids == 0, -100)
print('fwd', flush=True)
out = m(ids, attention_mask=am, labels=labels)
torch.cuda.synchronize(); print('loss', out.loss.item(), flush=True)
out.loss.backward()
torch.cuda.synchronize(); print('bwd ok', flush=True)
# graph capture of forward+backward
for p in m.parameters():
    p.grad = None
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        m.zero_grad(set_to_none=False)
        m(ids, attention_mask=am, labels=labels).loss.backward()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize(); print('warm ok', flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    lo = m(ids, attention_mask=am, labels=labels).loss
    lo.backward()
torch.cuda.synchronize(); print('captured', flush=True)
for i in range(3):
    g.replay(); torch.cuda.synchronize(); print('replay', i, lo.item(), flush=True)
