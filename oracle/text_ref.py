"""Oracle: text-conditioning stand-ins (tokenizer + CLIP text transformer).  TEST INFRASTRUCTURE ONLY.

Reference call sites: `tokenizer(prompts, padding='max_length', max_length=tokenizer.model_max_length,
truncation=True, return_tensors='pt').input_ids` and `text_encoder(ids)[0]`
(training/sid_sd_util.py:170-172, 221-240).  The real objects come from
transformers==4.40.1 + a downloaded vocabulary; no vocabulary or weights exist offline.

  * `HashTokenizerRef`: deterministic word -> id map with CLIP's BOS/EOS/pad structure.  It is
    NOT the CLIP BPE; it only has to give identical ids on both sides of a parity test.
  * `CLIPTextRef`: plain-torch restatement of the CLIP text transformer (causal mask, pre-LN,
    quick_gelu / gelu, final LayerNorm); pinned against `transformers.CLIPTextModel` (installed
    in this image) by weight sharing in tests/test_text_encoder.py.
"""
import hashlib
from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as F


class HashTokenizerRef:
    bos_token_id = 49406
    eos_token_id = 49407
    vocab_size = 49408

    def __init__(self, model_max_length=77, pad_token_id=49407):
        self.model_max_length = model_max_length
        self.pad_token_id = pad_token_id

    def _word_id(self, w):
        return int.from_bytes(hashlib.sha1(w.encode('utf-8')).digest()[:4], 'little') % 49406

    def __call__(self, text, padding='max_length', max_length=None, truncation=True, return_tensors='pt'):
        if isinstance(text, str):
            text = [text]
        L = max_length or self.model_max_length
        rows = []
        for s in text:
            ids = [self._word_id(w) for w in s.lower().split()][: L - 2]
            ids = [self.bos_token_id] + ids + [self.eos_token_id]
            ids = ids + [self.pad_token_id] * (L - len(ids))
            rows.append(ids)
        return SimpleNamespace(input_ids=torch.tensor(rows, dtype=torch.long))


class _Layer(nn.Module):
    def __init__(self, d, heads, dff, act):
        super().__init__()
        self.heads, self.act = heads, act
        self.layer_norm1 = nn.LayerNorm(d)
        self.q_proj, self.k_proj, self.v_proj, self.out_proj = (nn.Linear(d, d) for _ in range(4))
        self.layer_norm2 = nn.LayerNorm(d)
        self.fc1, self.fc2 = nn.Linear(d, dff), nn.Linear(dff, d)

    def forward(self, x, mask):
        B, L, D = x.shape
        h = self.layer_norm1(x)
        hd = D // self.heads
        q = self.q_proj(h).view(B, L, self.heads, hd).transpose(1, 2)
        k = self.k_proj(h).view(B, L, self.heads, hd).transpose(1, 2)
        v = self.v_proj(h).view(B, L, self.heads, hd).transpose(1, 2)
        s = q @ k.transpose(-1, -2) * hd ** -0.5 + mask
        o = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B, L, D)
        x = x + self.out_proj(o)
        h = self.fc1(self.layer_norm2(x))
        h = h * torch.sigmoid(1.702 * h) if self.act == 'quick_gelu' else F.gelu(h)
        return x + self.fc2(h)


class CLIPTextRef(nn.Module):
    def __init__(self, hidden=768, layers=12, heads=12, dff=3072, vocab=49408, max_pos=77, act='quick_gelu'):
        super().__init__()
        self.token_embedding = nn.Embedding(vocab, hidden)
        self.position_embedding = nn.Embedding(max_pos, hidden)
        self.layers = nn.ModuleList([_Layer(hidden, heads, dff, act) for _ in range(layers)])
        self.final_layer_norm = nn.LayerNorm(hidden)

    @property
    def device(self):
        return self.token_embedding.weight.device

    def forward(self, input_ids, attention_mask=None):
        B, L = input_ids.shape
        x = self.token_embedding(input_ids) + self.position_embedding.weight[:L]
        mask = torch.full((L, L), float('-inf'), device=x.device, dtype=x.dtype).triu(1)
        for lyr in self.layers:
            x = lyr(x, mask)
        return (self.final_layer_norm(x),)
