"""Text conditioning: tokenizer + CLIP text transformer ("stays PyTorch-ROCm" per BASELINE north_star).

Reference call sites: `tokenizer(prompts, padding='max_length', max_length=tokenizer.model_max_length,
truncation=True, return_tensors='pt').input_ids` and `text_encoder(ids)[0]`
(training/sid_sd_util.py:170-172, 221-240); objects come from transformers' CLIPTokenizer /
CLIPTextModel there (sid_sd_util.py:58-71).

  * `CLIPTextModel`: plain torch.nn CLIP text transformer with the transformers state_dict key names
    (`text_model.embeddings.token_embedding.weight`, `text_model.encoder.layers.N.self_attn.q_proj...`),
    so real SD text-encoder weights load with `load_state_dict`.  Checked against transformers'
    implementation by weight sharing in tests/test_text_encoder.py.
  * `HashTokenizer`: no CLIP vocabulary exists offline, so prompts are mapped to ids by a deterministic
    word hash with CLIP's BOS/EOS/padding structure.  It is a stand-in for throughput and parity work,
    NOT the CLIP BPE; `CLIPBPETokenizer.from_files(vocab.json, merges.txt)` is used when files are given.
  * `TextConditioner`: what the loop actually needs -- embeddings for a prompt batch and the
    loop-invariant "" embedding, which the reference recomputes 3x per iteration pair.
"""
import hashlib
import json
import os
import re
from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as F


class HashTokenizer:
    bos_token_id, eos_token_id, vocab_size = 49406, 49407, 49408

    def __init__(self, model_max_length=77, pad_token_id=49407):
        self.model_max_length, self.pad_token_id = model_max_length, pad_token_id

    def _encode(self, text):
        return [int.from_bytes(hashlib.sha1(w.encode('utf-8')).digest()[:4], 'little') % 49406 for w in text.lower().split()]

    def __call__(self, text, padding='max_length', max_length=None, truncation=True, return_tensors='pt'):
        if isinstance(text, str):
            text = [text]
        L = max_length or self.model_max_length
        rows = []
        for s in text:
            ids = [self.bos_token_id] + self._encode(s)[: L - 2] + [self.eos_token_id]
            rows.append(ids + [self.pad_token_id] * (L - len(ids)))
        return SimpleNamespace(input_ids=torch.tensor(rows, dtype=torch.long))


class CLIPBPETokenizer(HashTokenizer):
    """Byte-pair tokenizer driven by CLIP's vocab.json / merges.txt (lower-cased, whitespace-cleaned text)."""

    def __init__(self, vocab, merges, model_max_length=77, pad_token_id=49407):
        super().__init__(model_max_length, pad_token_id)
        self.vocab = vocab
        self.ranks = {tuple(m.split()): i for i, m in enumerate(merges)}
        self.pat = re.compile(r"<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[a-z]+|[0-9]|[^\sa-z0-9]+", re.I)
        self.cache = {}

    @classmethod
    def from_files(cls, vocab_json, merges_txt, **kw):
        with open(vocab_json) as f:
            vocab = json.load(f)
        with open(merges_txt) as f:
            merges = [ln.strip() for ln in f.read().split('\n')[1:] if ln.strip()]
        return cls(vocab, merges, **kw)

    def _bpe(self, token):
        if token in self.cache:
            return self.cache[token]
        word = tuple(token[:-1]) + (token[-1] + '</w>',)
        while len(word) > 1:
            pairs = {(a, b) for a, b in zip(word, word[1:])}
            best = min(pairs, key=lambda p: self.ranks.get(p, float('inf')))
            if best not in self.ranks:
                break
            out, i = [], 0
            while i < len(word):
                if i < len(word) - 1 and (word[i], word[i + 1]) == best:
                    out.append(word[i] + word[i + 1]); i += 2
                else:
                    out.append(word[i]); i += 1
            word = tuple(out)
        self.cache[token] = word
        return word

    def _encode(self, text):
        ids = []
        for tok in self.pat.findall(re.sub(r'\s+', ' ', text.strip()).lower()):
            ids.extend(self.vocab.get(t, self.eos_token_id) for t in self._bpe(tok))
        return ids


class _Attn(nn.Module):
    def __init__(self, d, heads):
        super().__init__()
        self.heads = heads
        self.k_proj, self.v_proj, self.q_proj, self.out_proj = (nn.Linear(d, d) for _ in range(4))

    def forward(self, h, mask):
        B, L, D = h.shape
        hd = D // self.heads
        q, k, v = (p(h).view(B, L, self.heads, hd).transpose(1, 2) for p in (self.q_proj, self.k_proj, self.v_proj))
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=mask)
        return self.out_proj(o.transpose(1, 2).reshape(B, L, D))


class _MLP(nn.Module):
    def __init__(self, d, dff, act):
        super().__init__()
        self.fc1, self.fc2, self.act = nn.Linear(d, dff), nn.Linear(dff, d), act

    def forward(self, x):
        h = self.fc1(x)
        h = h * torch.sigmoid(1.702 * h) if self.act == 'quick_gelu' else F.gelu(h)
        return self.fc2(h)


class _Layer(nn.Module):
    def __init__(self, d, heads, dff, act):
        super().__init__()
        self.self_attn = _Attn(d, heads)
        self.layer_norm1 = nn.LayerNorm(d)
        self.mlp = _MLP(d, dff, act)
        self.layer_norm2 = nn.LayerNorm(d)

    def forward(self, x, mask):
        x = x + self.self_attn(self.layer_norm1(x), mask)
        return x + self.mlp(self.layer_norm2(x))


class _Embeddings(nn.Module):
    def __init__(self, vocab, d, max_pos):
        super().__init__()
        self.token_embedding = nn.Embedding(vocab, d)
        self.position_embedding = nn.Embedding(max_pos, d)


class _Encoder(nn.Module):
    def __init__(self, d, layers, heads, dff, act):
        super().__init__()
        self.layers = nn.ModuleList([_Layer(d, heads, dff, act) for _ in range(layers)])


class _TextTransformer(nn.Module):
    def __init__(self, d, layers, heads, dff, vocab, max_pos, act):
        super().__init__()
        self.embeddings = _Embeddings(vocab, d, max_pos)
        self.encoder = _Encoder(d, layers, heads, dff, act)
        self.final_layer_norm = nn.LayerNorm(d)


class CLIPTextModel(nn.Module):
    def __init__(self, hidden=768, layers=12, heads=12, dff=3072, vocab=49408, max_pos=77, act='quick_gelu'):
        super().__init__()
        self.text_model = _TextTransformer(hidden, layers, heads, dff, vocab, max_pos, act)
        self.config = SimpleNamespace(hidden_size=hidden, num_hidden_layers=layers, max_position_embeddings=max_pos)

    @property
    def device(self):
        return self.text_model.embeddings.token_embedding.weight.device

    @property
    def dtype(self):
        return self.text_model.embeddings.token_embedding.weight.dtype

    def forward(self, input_ids, attention_mask=None):
        tm = self.text_model
        B, L = input_ids.shape
        x = tm.embeddings.token_embedding(input_ids) + tm.embeddings.position_embedding.weight[:L]
        mask = torch.full((L, L), float('-inf'), device=x.device, dtype=x.dtype).triu(1)
        for lyr in tm.encoder.layers:
            x = lyr(x, mask)
        return (tm.final_layer_norm(x),)


TEXT_CONFIGS = {
    'sd15': dict(hidden=768, layers=12, heads=12, dff=3072, act='quick_gelu'),            # CLIP ViT-L/14 text
    'sd21-base': dict(hidden=1024, layers=23, heads=16, dff=4096, act='gelu'),            # OpenCLIP-H text (penultimate)
}


class TextConditioner:
    """prompts -> bf16 [B, L, D] text states on the GPU, with the constant "" state cached."""

    def __init__(self, tokenizer, text_encoder, out_dtype=torch.bfloat16):
        self.tokenizer, self.text_encoder, self.out_dtype = tokenizer, text_encoder, out_dtype
        self._uncond = None

    @torch.no_grad()
    def encode(self, prompts):
        ids = self.tokenizer(list(prompts), padding='max_length', max_length=self.tokenizer.model_max_length, truncation=True,
                             return_tensors='pt').input_ids
        dev = self.text_encoder.device
        if dev.type == 'cuda':           # pinned + non-blocking: a pageable copy would block the host until the stream drains
            ids = ids.pin_memory().to(dev, non_blocking=True)
        else:
            ids = ids.to(dev)
        return self.text_encoder(ids)[0].to(self.out_dtype).contiguous()

    def uncond(self, batch):
        if self._uncond is None:
            self._uncond = self.encode([''])
        return self._uncond.expand(batch, -1, -1).contiguous()
