"""Throughput of the BERT text encoder forward at the hot path's shape (development aid / DESIGN.md numbers).
Usage: python tools/bench_text_encoder.py [samples=16] [tokens=40] [layers=12] [heads=4]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from layoutdetr_amd.training import med
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
T = int(sys.argv[2]) if len(sys.argv) > 2 else 40
L = int(sys.argv[3]) if len(sys.argv) > 3 else 12
H = int(sys.argv[4]) if len(sys.argv) > 4 else 4
cfg = med.BertConfig(num_hidden_layers=L, num_attention_heads=H)
m = med.BertModel(cfg).eval().requires_grad_(False).to(dev)
S = B * 9
ids = torch.randint(1, 30000, (S, T), device=dev); am = torch.ones(S, T, dtype=torch.long, device=dev)
f = lambda: m(ids, attention_mask=am)
f(); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    f()
g.replay(); torch.cuda.synchronize()
s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(5): g.replay()
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 5
d, I = cfg.hidden_size, cfg.intermediate_size
flops = L * (2.0 * S * T * (4 * d * d + 2 * d * I) + 4.0 * S * T * T * d)
print(f'{S} texts x {T} tokens, {L} layers, {H} heads x {d // H}: {ms:.2f} ms per forward, {S * T / ms * 1e3:.0f} tokens/s, {flops / ms / 1e9:.1f} TFLOP/s')
