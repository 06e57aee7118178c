"""CPU restatement of the reference's BERT text encoder in text mode (TEST INFRASTRUCTURE — only tests/, smoke() and bench.py's
cpu_baseline may import this package; the product path never does).

Follows training/med.py of the reference: BertEmbeddings.forward :74-97 (word + absolute position embeddings, LayerNorm),
BertSelfAttention.forward :139-228 (QK^T / sqrt(dh) + additive mask, softmax, PV; self-attention branch), BertSelfOutput
:231-241, BertIntermediate :296-307 (erf GELU), BertOutput :310-320, BertLayer.forward :337-380 with mode='text' (no
cross-attention), BertEncoder.forward :404-486, and BertModel.forward's mask preparation :709-748
(extended mask = (1 - attention_mask) * -10000).  Dropout off (eval).  Plain torch CPU fp32 ops on a state_dict with the
reference's parameter names.  Pinned by tests/golden/bert_text*.npz (oracle/gen_golden.py gen_bert: outputs of the reference's
own BertEmbeddings + BertEncoder modules imported from /root/reference)."""
import math

import torch
import torch.nn.functional as F


def bert_text_forward(sd, num_heads, input_ids, attention_mask, eps=1e-12):
    """sd: {'embeddings.word_embeddings.weight', ..., 'encoder.layer.N....'}; returns last_hidden_state [B, T, hidden]."""
    B, T = input_ids.shape
    x = sd['embeddings.word_embeddings.weight'][input_ids] + sd['embeddings.position_embeddings.weight'][:T][None]
    d = x.shape[-1]
    x = F.layer_norm(x, (d,), sd['embeddings.LayerNorm.weight'], sd['embeddings.LayerNorm.bias'], eps)
    ext = (1.0 - attention_mask[:, None, None, :].to(torch.float32)) * -10000.0
    dh = d // num_heads
    n_layers = 1 + max(int(k.split('.')[2]) for k in sd if k.startswith('encoder.layer.'))
    for i in range(n_layers):
        p = f'encoder.layer.{i}.'
        lin = lambda t, name: t @ sd[p + name + '.weight'].t() + sd[p + name + '.bias']
        heads = lambda t: t.reshape(B, T, num_heads, dh).permute(0, 2, 1, 3)
        q, k, v = heads(lin(x, 'attention.self.query')), heads(lin(x, 'attention.self.key')), heads(lin(x, 'attention.self.value'))
        scores = q @ k.transpose(-1, -2) / math.sqrt(dh) + ext
        ctx = (torch.softmax(scores, dim=-1) @ v).permute(0, 2, 1, 3).reshape(B, T, d)
        a = lin(ctx, 'attention.output.dense')
        x = F.layer_norm(a + x, (d,), sd[p + 'attention.output.LayerNorm.weight'], sd[p + 'attention.output.LayerNorm.bias'], eps)
        h = lin(x, 'intermediate.dense')
        h = h * 0.5 * (1.0 + torch.erf(h / math.sqrt(2.0)))
        f = lin(h, 'output.dense')
        x = F.layer_norm(f + x, (d,), sd[p + 'output.LayerNorm.weight'], sd[p + 'output.LayerNorm.bias'], eps)
    return x


def bert_lm_loss(sd, num_heads, input_ids, attention_mask, labels, eps=1e-12):
    """Text-mode BertLMHeadModel (reference training/med.py: BertModel.get_extended_attention_mask with is_decoder :704-739 ->
    causal * padding mask; BertLayer mode='text' :337-380 (no cross-attention); BertOnlyMLMHead :504-545 with the decoder weight
    tied to the word embeddings; shifted label-smoothed cross entropy :911-916).  sd keys: 'bert.embeddings.*',
    'bert.encoder.layer.N.*', 'cls.predictions.*' (decoder.weight optional: tied).  Returns (loss, logits[:, :-1])."""
    B, T = input_ids.shape
    W = sd['bert.embeddings.word_embeddings.weight']
    x = W[input_ids] + sd['bert.embeddings.position_embeddings.weight'][:T][None]
    d = x.shape[-1]
    x = F.layer_norm(x, (d,), sd['bert.embeddings.LayerNorm.weight'], sd['bert.embeddings.LayerNorm.bias'], eps)
    causal = torch.tril(torch.ones(T, T))[None, None]
    ext = (1.0 - causal * attention_mask[:, None, None, :].to(torch.float32)) * -10000.0
    dh = d // num_heads
    n_layers = 1 + max(int(k.split('.')[3]) for k in sd if k.startswith('bert.encoder.layer.'))
    for i in range(n_layers):
        p = f'bert.encoder.layer.{i}.'
        lin = lambda t, name: t @ sd[p + name + '.weight'].t() + sd[p + name + '.bias']
        heads = lambda t: t.reshape(B, T, num_heads, dh).permute(0, 2, 1, 3)
        q, k, v = heads(lin(x, 'attention.self.query')), heads(lin(x, 'attention.self.key')), heads(lin(x, 'attention.self.value'))
        ctx = (torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(dh) + ext, dim=-1) @ v).permute(0, 2, 1, 3).reshape(B, T, d)
        x = F.layer_norm(lin(ctx, 'attention.output.dense') + x, (d,), sd[p + 'attention.output.LayerNorm.weight'], sd[p + 'attention.output.LayerNorm.bias'], eps)
        h = lin(x, 'intermediate.dense')
        h = h * 0.5 * (1.0 + torch.erf(h / math.sqrt(2.0)))
        x = F.layer_norm(lin(h, 'output.dense') + x, (d,), sd[p + 'output.LayerNorm.weight'], sd[p + 'output.LayerNorm.bias'], eps)
    t = x @ sd['cls.predictions.transform.dense.weight'].t() + sd['cls.predictions.transform.dense.bias']
    t = t * 0.5 * (1.0 + torch.erf(t / math.sqrt(2.0)))
    t = F.layer_norm(t, (d,), sd['cls.predictions.transform.LayerNorm.weight'], sd['cls.predictions.transform.LayerNorm.bias'], eps)
    logits = t @ W.t() + sd['cls.predictions.bias']
    sh = logits[:, :-1].contiguous()
    loss = F.cross_entropy(sh.view(-1, sh.shape[-1]), labels[:, 1:].reshape(-1), ignore_index=-100, label_smoothing=0.1)
    return loss, sh
