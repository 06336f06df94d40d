"""ORACLE (test infrastructure — never imported by the product path under wise_amd/).

CPU fp32 restatement of the text path the reference runs at /root/reference/src/feature/mlfoundation_openclip.py:103-108
for its DEFAULT model pair, xlm-roberta-large-ViT-H-14/frozen_laion5b_s13b_b90k (extract-features.py:192):

    text_features = self.model.encode_text(self.tokenizer(text_query))
    text_features /= text_features.norm(dim=-1, keepdim=True)

For that model `encode_text` is open_clip_torch==2.24.0 hf_model.py `HFTextEncoder.forward` (requirements.txt:11; not
vendored): attn_mask = (x != pad_token_id); out = XLMRobertaModel(input_ids=x, attention_mask=attn_mask);
pooled = MeanPooler (sum of masked last_hidden_state / sum of the mask); projected = proj(pooled) with proj =
Linear(W, (W+D)/2, no bias) -> GELU -> Linear((W+D)/2, D, no bias).  XLMRobertaModel = BERT-style POST-LN encoder with
word + position + token-type embeddings, position ids = cumsum(mask) * mask + padding_idx (transformers
modeling_xlm_roberta.py create_position_ids_from_input_ids), embedding LayerNorm, erf GELU, layer_norm_eps 1e-5.

PINNING: pinned against transformers' XLMRobertaModel (in the container) fed the same seeded weights
(oracle/make_golden_xlmr.py, max |diff| ~1e-6 on hidden states); the pooler and the projection are the four lines
above.  Parity with the real laion5b checkpoint / the real sentencepiece vocabulary: UNPINNED offline.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch

from .vit_ref import gelu, layer_norm


def xlmr_text_forward(sd: Dict[str, torch.Tensor], tokens: torch.Tensor, *, heads: int, pad_id: int = 1,
                      taps: Optional[List[torch.Tensor]] = None, normalize: bool = True) -> torch.Tensor:
    """tokens int [B,T] -> [B,D] fp32.  taps (if a list) receives the hidden states after the embeddings and every layer."""
    tok = tokens.to(torch.int64)
    B, T = tok.shape
    e = "text.transformer.embeddings."
    mask = (tok != pad_id)
    pos = torch.cumsum(mask.to(torch.int64), dim=1) * mask.to(torch.int64) + pad_id
    x = (sd[e + "word_embeddings.weight"].to(torch.float32)[tok] + sd[e + "position_embeddings.weight"].to(torch.float32)[pos]
         + sd[e + "token_type_embeddings.weight"].to(torch.float32)[0])
    x = layer_norm(x, sd[e + "LayerNorm.weight"], sd[e + "LayerNorm.bias"])
    if taps is not None:
        taps.append(x.clone())
    Wd = x.shape[-1]
    dh = Wd // heads
    neg = torch.zeros(B, 1, 1, T)
    neg.masked_fill_(~mask[:, None, None, :], float("-inf"))      # padded keys are invisible to every query
    n_layers = 0
    while f"text.transformer.encoder.layer.{n_layers}.attention.self.query.weight" in sd:
        n_layers += 1
    for i in range(n_layers):
        p = f"text.transformer.encoder.layer.{i}."
        q = x @ sd[p + "attention.self.query.weight"].t() + sd[p + "attention.self.query.bias"]
        k = x @ sd[p + "attention.self.key.weight"].t() + sd[p + "attention.self.key.bias"]
        v = x @ sd[p + "attention.self.value.weight"].t() + sd[p + "attention.self.value.bias"]
        q = q.reshape(B, T, heads, dh).transpose(1, 2)
        k = k.reshape(B, T, heads, dh).transpose(1, 2)
        v = v.reshape(B, T, heads, dh).transpose(1, 2)
        s = (q @ k.transpose(-1, -2)) / math.sqrt(dh) + neg
        s = s - s.max(dim=-1, keepdim=True).values
        ex = torch.exp(s)
        pr = ex / ex.sum(dim=-1, keepdim=True)
        o = (pr @ v).transpose(1, 2).reshape(B, T, Wd)
        a = o @ sd[p + "attention.output.dense.weight"].t() + sd[p + "attention.output.dense.bias"]
        x = layer_norm(a + x, sd[p + "attention.output.LayerNorm.weight"], sd[p + "attention.output.LayerNorm.bias"])
        h = gelu(x @ sd[p + "intermediate.dense.weight"].t() + sd[p + "intermediate.dense.bias"])
        f = h @ sd[p + "output.dense.weight"].t() + sd[p + "output.dense.bias"]
        x = layer_norm(f + x, sd[p + "output.LayerNorm.weight"], sd[p + "output.LayerNorm.bias"])
        if taps is not None:
            taps.append(x.clone())
    m = mask.to(torch.float32)
    pooled = (x * m[:, :, None]).sum(dim=1) / m.sum(dim=1, keepdim=True)
    out = gelu(pooled @ sd["text.proj.0.weight"].t()) @ sd["text.proj.2.weight"].t()
    if normalize:
        out = out / torch.linalg.norm(out, dim=-1, keepdim=True)
    return out
