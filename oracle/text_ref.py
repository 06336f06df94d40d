"""ORACLE (test infrastructure — never imported by the product path under wise_amd/).

CPU fp32 restatement of the text path the reference runs at
/root/reference/src/feature/mlfoundation_openclip.py:103-108 (reached from FeatureSearchIndex.search,
src/index/feature_search_index.py:112):

    text_features = self.model.encode_text(self.tokenizer(text_query))
    text_features /= text_features.norm(dim=-1, keepdim=True)

`encode_text` lives in the un-vendored dependency open_clip_torch==2.24.0 (requirements.txt:11); its published
definition is restated here: x = token_embedding[text] + positional_embedding -> L pre-LN residual blocks with a
causal (upper-triangular -inf) attention mask -> ln_final -> x[b, text[b].argmax()] @ text_projection.

PINNING: pinned against transformers' CLIPTextModelWithProjection (an independent implementation that IS in the
container) fed the same seeded weights, with eos_token_id=2 so that it pools at argmax(input_ids) exactly like
open_clip (oracle/make_golden_text.py, max |diff| ~1e-6).  Parity with a real open_clip checkpoint: UNPINNED offline.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch

from .vit_ref import gelu, layer_norm, quick_gelu


def text_forward(sd: Dict[str, torch.Tensor], tokens: torch.Tensor, *, heads: int, act: str = "quick_gelu",
                 taps: Optional[List[torch.Tensor]] = None, normalize: bool = True) -> torch.Tensor:
    """tokens int [B,T] -> [B,D] fp32.  taps (if a list) receives the residual stream after every block."""
    tok = tokens.to(torch.int64)
    B, T = tok.shape
    x = sd["token_embedding.weight"].to(torch.float32)[tok] + sd["positional_embedding"].to(torch.float32)[:T]
    Wd = x.shape[-1]
    dh = Wd // heads
    mask = torch.full((T, T), float("-inf")).triu(1)  # query t sees keys <= t
    actf = quick_gelu if act == "quick_gelu" else gelu
    n_layers = 0
    while f"transformer.resblocks.{n_layers}.ln_1.weight" in sd:
        n_layers += 1
    for i in range(n_layers):
        p = f"transformer.resblocks.{i}."
        h = layer_norm(x, sd[p + "ln_1.weight"], sd[p + "ln_1.bias"])
        qkv = h @ sd[p + "attn.in_proj_weight"].t() + sd[p + "attn.in_proj_bias"]
        q, k, v = qkv.split(Wd, dim=-1)
        q = q.reshape(B, T, heads, dh).transpose(1, 2)
        k = k.reshape(B, T, heads, dh).transpose(1, 2)
        v = v.reshape(B, T, heads, dh).transpose(1, 2)
        s = (q @ k.transpose(-1, -2)) / math.sqrt(dh) + mask
        s = s - s.max(dim=-1, keepdim=True).values
        e = torch.exp(s)
        pr = e / e.sum(dim=-1, keepdim=True)
        o = (pr @ v).transpose(1, 2).reshape(B, T, Wd)
        x = x + o @ sd[p + "attn.out_proj.weight"].t() + sd[p + "attn.out_proj.bias"]
        h = layer_norm(x, sd[p + "ln_2.weight"], sd[p + "ln_2.bias"])
        h = actf(h @ sd[p + "mlp.c_fc.weight"].t() + sd[p + "mlp.c_fc.bias"])
        x = x + h @ sd[p + "mlp.c_proj.weight"].t() + sd[p + "mlp.c_proj.bias"]
        if taps is not None:
            taps.append(x.clone())
    x = layer_norm(x, sd["ln_final.weight"], sd["ln_final.bias"])
    pooled = x[torch.arange(B), tok.argmax(dim=-1)]
    out = pooled @ sd["text_projection"].to(torch.float32)
    if normalize:
        out = out / out.norm(dim=-1, keepdim=True)
    return out


def attention_causal_ref(qkv: torch.Tensor, B: int, T: int, H: int) -> torch.Tensor:
    """qkv [B*T, 3*H*64] -> o [B*T, H*64] with the causal mask; the op the text tower's attention computes."""
    Wd = H * 64
    q, k, v = qkv.to(torch.float32).reshape(B, T, 3 * Wd).split(Wd, dim=-1)
    q = q.reshape(B, T, H, 64).transpose(1, 2)
    k = k.reshape(B, T, H, 64).transpose(1, 2)
    v = v.reshape(B, T, H, 64).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) / 8.0 + torch.full((T, T), float("-inf")).triu(1)
    return (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(B * T, Wd)
