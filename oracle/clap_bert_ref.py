"""ORACLE (test infrastructure — never imported by the product path under wise_amd/).

CPU fp32 restatement of the text path of MS-CLAP version '2022' — what the reference runs at
/root/reference/src/feature/microsoft_clap.py:53-58 when the feature id is `microsoft/clap/2022/...`:

    preprocessed_text = self.model.preprocess_text(text)
    text_embeddings = self.model.clap.caption_encoder(preprocessed_text)
    text_embeddings = text_embeddings / torch.norm(text_embeddings, dim=-1, keepdim=True)

msclap==1.3.3 (requirements.txt:25-26; not vendored), config_2022: text_model 'bert-base-uncased', text_len 100,
transformer_embed_dim 768, d_proj 1024.  `caption_encoder` = TextEncoder: out = BertModel(**tokens)[0][:, 0, :] (the [CLS]
row of the last hidden state — not BERT's tanh pooler), then Projection(768, 1024): e1 = linear1(x), e2 = linear2(gelu(e1))
(dropout is identity in eval), LayerNorm(e1 + e2), both Linear layers without bias.  BertModel = POST-LN encoder with
word + absolute position (0..T-1) + token-type (all zero) embeddings, embedding LayerNorm, erf GELU, layer_norm_eps 1e-12,
attention mask from the tokenizer's padding (id 0).

PINNING: the encoder is pinned against transformers' BertModel (in the container) on the same seeded weights
(oracle/make_golden_clap_bert.py, max |diff| ~1e-5 on the [CLS] row); the CLS pooling and the Projection are the lines
above (the Projection is the one of the 2023 model, oracle/htsat_ref.py).  Parity with the real CLAP 2022 checkpoint and
the real WordPiece vocabulary: UNPINNED offline.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch

from .vit_ref import gelu, layer_norm

EPS = 1e-12


def bert_hidden(sd: Dict[str, torch.Tensor], tokens: torch.Tensor, *, heads: int, pad_id: int = 0,
                taps: Optional[List[torch.Tensor]] = None) -> torch.Tensor:
    """tokens int [B, T] -> last hidden state [B, T, W] fp32 (keys as in transformers' BertModel under `base.`)."""
    tok = tokens.to(torch.int64)
    B, T = tok.shape
    e = "base.embeddings."
    mask = tok != pad_id
    x = (sd[e + "word_embeddings.weight"].float()[tok] + sd[e + "position_embeddings.weight"].float()[:T][None]
         + sd[e + "token_type_embeddings.weight"].float()[0])
    x = layer_norm(x, sd[e + "LayerNorm.weight"], sd[e + "LayerNorm.bias"], eps=EPS)
    if taps is not None:
        taps.append(x.clone())
    Wd = x.shape[-1]
    dh = Wd // heads
    neg = torch.zeros(B, 1, 1, T)
    neg.masked_fill_(~mask[:, None, None, :], float("-inf"))
    n_layers = 0
    while f"base.encoder.layer.{n_layers}.attention.self.query.weight" in sd:
        n_layers += 1
    for i in range(n_layers):
        p = f"base.encoder.layer.{i}."
        q = x @ sd[p + "attention.self.query.weight"].t() + sd[p + "attention.self.query.bias"]
        k = x @ sd[p + "attention.self.key.weight"].t() + sd[p + "attention.self.key.bias"]
        v = x @ sd[p + "attention.self.value.weight"].t() + sd[p + "attention.self.value.bias"]
        q = q.reshape(B, T, heads, dh).transpose(1, 2)
        k = k.reshape(B, T, heads, dh).transpose(1, 2)
        v = v.reshape(B, T, heads, dh).transpose(1, 2)
        s = (q @ k.transpose(-1, -2)) / math.sqrt(dh) + neg
        pr = torch.softmax(s, dim=-1)
        o = (pr @ v).transpose(1, 2).reshape(B, T, Wd)
        a = o @ sd[p + "attention.output.dense.weight"].t() + sd[p + "attention.output.dense.bias"]
        x = layer_norm(a + x, sd[p + "attention.output.LayerNorm.weight"], sd[p + "attention.output.LayerNorm.bias"], eps=EPS)
        h = gelu(x @ sd[p + "intermediate.dense.weight"].t() + sd[p + "intermediate.dense.bias"])
        f = h @ sd[p + "output.dense.weight"].t() + sd[p + "output.dense.bias"]
        x = layer_norm(f + x, sd[p + "output.LayerNorm.weight"], sd[p + "output.LayerNorm.bias"], eps=EPS)
        if taps is not None:
            taps.append(x.clone())
    return x


def caption_forward_2022(sd: Dict[str, torch.Tensor], tokens: torch.Tensor, *, heads: int = 12,
                         normalize: bool = True) -> torch.Tensor:
    """the reference's extract_text_features for version '2022' on token ids: [B, T] -> [B, 1024]"""
    with torch.no_grad():
        cls = bert_hidden(sd, tokens, heads=heads)[:, 0, :]
        e1 = cls @ sd["projection.linear1.weight"].t()
        e2 = gelu(e1) @ sd["projection.linear2.weight"].t()
        out = layer_norm(e1 + e2, sd["projection.layer_norm.weight"], sd["projection.layer_norm.bias"])
        if normalize:
            out = out / torch.linalg.norm(out, dim=-1, keepdim=True)
        return out
