"""Host-side BERT WordPiece tokenizer adapter: element text strings -> `TextTokens` (the boundary input of the text path).

The reference tokenises inside Generator/Discriminator.forward with `transformers.BertTokenizer.from_pretrained('bert-base-uncased')`
plus two added tokens (training/blip.py:190-195, networks_detr.py:145,289: `padding='max_length', truncation=True,
max_length=max_text_length`).  That constructor downloads the vocabulary; here the same arithmetic (uncased basic tokenisation +
greedy longest-match WordPiece, as published with BERT) runs from a LOCAL `vocab.txt`, with no `transformers` dependency on the
hot path.  tests/test_host_cpu.py checks it id-for-id against `transformers.BertTokenizer` built from the same file.
"""
import unicodedata

import torch


def _is_punct(ch):
    cp = ord(ch)
    if 33 <= cp <= 47 or 58 <= cp <= 64 or 91 <= cp <= 96 or 123 <= cp <= 126:
        return True
    return unicodedata.category(ch).startswith('P')


def _is_cjk(cp):
    return (0x4E00 <= cp <= 0x9FFF or 0x3400 <= cp <= 0x4DBF or 0x20000 <= cp <= 0x2A6DF or 0x2A700 <= cp <= 0x2B73F or
            0x2B740 <= cp <= 0x2B81F or 0x2B820 <= cp <= 0x2CEAF or 0xF900 <= cp <= 0xFAFF or 0x2F800 <= cp <= 0x2FA1F)


class BertWordPieceTokenizer(object):
    def __init__(self, vocab_file, max_word_chars=100):
        with open(vocab_file, encoding='utf-8') as fh:
            toks = [line.rstrip('\n') for line in fh]
        self.vocab = {t: i for i, t in enumerate(toks)}
        self.unk, self.cls, self.sep, self.pad = (self.vocab[t] for t in ('[UNK]', '[CLS]', '[SEP]', '[PAD]'))
        self.pad_token_id = self.pad
        # blip.init_tokenizer: add_special_tokens bos '[DEC]' then '[ENC]' -> the two ids after the base vocabulary
        self.bos_token_id = self.vocab.get('[DEC]', len(toks))
        self.enc_token_id = self.vocab.get('[ENC]', self.bos_token_id + 1)
        self.max_word_chars = max_word_chars

    def __len__(self):
        return max(len(self.vocab), self.enc_token_id + 1)

    def _basic(self, text):
        out = []
        for ch in text:
            cp = ord(ch)
            if cp == 0 or cp == 0xFFFD or (unicodedata.category(ch).startswith('C') and ch not in '\t\n\r'):
                continue
            if ch in ' \t\n\r' or unicodedata.category(ch) == 'Zs':
                out.append(' ')
            elif _is_cjk(cp):
                out.append(f' {ch} ')
            else:
                out.append(ch)
        words = []
        for w in ''.join(out).split():
            w = ''.join(c for c in unicodedata.normalize('NFD', w.lower()) if unicodedata.category(c) != 'Mn')
            cur = ''
            for ch in w:
                if _is_punct(ch):
                    if cur:
                        words.append(cur); cur = ''
                    words.append(ch)
                else:
                    cur += ch
            if cur:
                words.append(cur)
        return words

    def _wordpiece(self, word):
        if len(word) > self.max_word_chars:
            return [self.unk]
        ids, start = [], 0
        while start < len(word):
            end, cur = len(word), None
            while start < end:
                piece = ('##' if start > 0 else '') + word[start:end]
                if piece in self.vocab:
                    cur = self.vocab[piece]; break
                end -= 1
            if cur is None:
                return [self.unk]
            ids.append(cur); start = end
        return ids

    def encode(self, text, max_length):
        ids = [i for w in self._basic(text) for i in self._wordpiece(w)][:max_length - 2]
        ids = [self.cls] + ids + [self.sep]
        return ids + [self.pad] * (max_length - len(ids)), [1] * len(ids) + [0] * (max_length - len(ids))

    def __call__(self, texts, padding='max_length', truncation=True, max_length=256, return_tensors='pt'):
        enc = [self.encode(t, max_length) for t in texts]
        return torch.tensor([e[0] for e in enc], dtype=torch.int64), torch.tensor([e[1] for e in enc], dtype=torch.int64)


def texts_to_tokens(tokenizer, bbox_text, max_text_length, device=None, trim=True):
    """bbox_text: list (batch) of lists (elements) of strings, as the reference's loader hands them over (training_loop.py:246-247)
    -> TextTokens(input_ids [B,N,T], attention_mask [B,N,T], text_len [B,N] = character counts, networks_detr.py:149).
    trim: drop the all-padding tail columns shared by the whole batch (T = longest text instead of max_text_length): padded
    positions are masked out of every attention, so the text encoder's CLS feature is unchanged and the LM loss ignores them."""
    from .networks_detr import TextTokens
    B, N = len(bbox_text), len(bbox_text[0])
    flat = [t for row in bbox_text for t in row]
    ids, am = tokenizer(flat, max_length=max_text_length)
    if trim:
        T = max(int(am.sum(1).max().item()), 2)
        ids, am = ids[:, :T], am[:, :T]
    tl = torch.tensor([len(t) for t in flat], dtype=torch.int64).clamp_(max=max_text_length - 1)
    tok = TextTokens(ids.view(B, N, -1), am.view(B, N, -1), tl.view(B, N), bos_token_id=tokenizer.bos_token_id, pad_token_id=tokenizer.pad_token_id)
    if device is not None:
        tok = TextTokens(tok.input_ids.to(device), tok.attention_mask.to(device), tok.text_len.to(device), tok.bos_token_id, tok.pad_token_id)
    return tok
