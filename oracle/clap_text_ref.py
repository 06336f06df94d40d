"""ORACLE (test infrastructure — never imported by the product path under wise_amd/).

CPU fp32 restatement of the caption path the reference runs at
/root/reference/src/feature/microsoft_clap.py:53-58:

    text_embeddings = self.model.clap.caption_encoder(preprocessed_text)
    text_embeddings = text_embeddings / torch.norm(text_embeddings, dim=-1, keepdim=True)

`caption_encoder` lives in the un-vendored msclap==1.3.3 (requirements.txt:25-26): `TextEncoder` = Hugging Face
GPT2Model (wte + wpe, pre-LN blocks with a causal mask, gelu_new MLP, ln_f), the hidden state at index
`ne(input_ids, 0).sum(-1) - 1`, then msclap `Projection`: e1 = linear1(x); e2 = linear2(gelu(e1)) (dropout off at
eval); LayerNorm(e1 + e2).

PINNING: the GPT-2 body is pinned against transformers' GPT2Model (in the container) fed the same seeded weights
(oracle/make_golden_clap_text.py, max |diff| ~1e-5); the Projection is the one restated for the audio head
(oracle/htsat_ref.py).  Parity with the real msclap checkpoint: UNPINNED offline.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch

from .vit_ref import gelu, layer_norm


def gelu_new(x: torch.Tensor) -> torch.Tensor:
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x ** 3)))


def gpt2_hidden(sd: Dict[str, torch.Tensor], tokens: torch.Tensor, heads: int,
                taps: Optional[List[torch.Tensor]] = None) -> torch.Tensor:
    """tokens int [B,T] -> last hidden state [B,T,W] after ln_f."""
    tok = tokens.to(torch.int64)
    B, T = tok.shape
    x = sd["base.wte.weight"][tok] + sd["base.wpe.weight"][:T]
    Wd = x.shape[-1]
    dh = Wd // heads
    mask = torch.full((T, T), float("-inf")).triu(1)
    n_layers = 0
    while f"base.h.{n_layers}.ln_1.weight" in sd:
        n_layers += 1
    for i in range(n_layers):
        p = f"base.h.{i}."
        h = layer_norm(x, sd[p + "ln_1.weight"], sd[p + "ln_1.bias"])
        qkv = h @ sd[p + "attn.c_attn.weight"] + sd[p + "attn.c_attn.bias"]   # Conv1D: x @ W[in,out] + b
        q, k, v = qkv.split(Wd, dim=-1)
        q = q.reshape(B, T, heads, dh).transpose(1, 2)
        k = k.reshape(B, T, heads, dh).transpose(1, 2)
        v = v.reshape(B, T, heads, dh).transpose(1, 2)
        s = (q @ k.transpose(-1, -2)) / math.sqrt(dh) + mask
        pr = torch.softmax(s, dim=-1)
        o = (pr @ v).transpose(1, 2).reshape(B, T, Wd)
        x = x + o @ sd[p + "attn.c_proj.weight"] + sd[p + "attn.c_proj.bias"]
        h = layer_norm(x, sd[p + "ln_2.weight"], sd[p + "ln_2.bias"])
        h = gelu_new(h @ sd[p + "mlp.c_fc.weight"] + sd[p + "mlp.c_fc.bias"])
        x = x + h @ sd[p + "mlp.c_proj.weight"] + sd[p + "mlp.c_proj.bias"]
        if taps is not None:
            taps.append(x.clone())
    return layer_norm(x, sd["base.ln_f.weight"], sd["base.ln_f.bias"])


def projection(sd: Dict[str, torch.Tensor], x: torch.Tensor) -> torch.Tensor:
    e1 = x @ sd["projection.linear1.weight"].t()
    e2 = gelu(e1) @ sd["projection.linear2.weight"].t()
    return layer_norm(e1 + e2, sd["projection.layer_norm.weight"], sd["projection.layer_norm.bias"])


def caption_forward(sd: Dict[str, torch.Tensor], tokens: torch.Tensor, heads: int = 12,
                    normalize: bool = True) -> torch.Tensor:
    """tokens int [B,T] (right-padded with 0) -> [B,1024] fp32."""
    hid = gpt2_hidden(sd, tokens, heads)
    B = tokens.shape[0]
    last = (tokens != 0).sum(-1) - 1
    out = projection(sd, hid[torch.arange(B), last])
    if normalize:
        out = out / torch.norm(out, dim=-1, keepdim=True)
    return out
