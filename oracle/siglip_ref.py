"""ORACLE (test infrastructure — never imported by the product path under wise_amd/).

CPU fp32 restatement of the two towers of open_clip's SigLIP models — `ViT-L-16-SigLIP-384/webli` is what the reference's
own end-to-end test extracts video features with (/root/reference/tests/test-kinetics-6.sh:91), through the same two
call sites as every open_clip model: src/feature/mlfoundation_openclip.py:99-100 (encode_image + L2 normalise) and
:105-107 (encode_text + L2 normalise).

Image tower = timm VisionTransformer as open_clip's `TimmModel(pool='map', proj='none')` builds it (neither timm nor
open_clip is vendored or installed): patch embedding with bias, learned positions, no class token, no pre-norm, pre-LN
blocks with LayerNorm eps 1e-6, final norm, AttentionPoolLatent (one latent query; q, kv, proj Linear layers;
x = x + mlp(norm(x)); token 0), no projection.  Text tower = open_clip TextTransformer with no_causal_mask, eps 1e-6,
pool_type 'last', text_projection = Linear with bias.

PINNING: pinned against transformers' SiglipVisionModel / SiglipTextModel (an independent port of the same
architecture, in the container) fed the same seeded weights (oracle/make_golden_siglip.py).  Parity with the webli
checkpoints, the timm key names the packer reads, and whether an installation's timm uses erf or tanh GELU: UNPINNED
offline (both activations are restated; `act` selects).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch

from .vit_ref import gelu, layer_norm


def gelu_tanh(x: torch.Tensor) -> torch.Tensor:
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x ** 3)))


def _attention(q, k, v, heads: int):
    B, Tq, Wd = q.shape
    Tk = k.shape[1]
    dh = Wd // heads
    q = q.reshape(B, Tq, heads, dh).transpose(1, 2)
    k = k.reshape(B, Tk, heads, dh).transpose(1, 2)
    v = v.reshape(B, Tk, heads, dh).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) / math.sqrt(dh)
    s = s - s.max(dim=-1, keepdim=True).values
    e = torch.exp(s)
    pr = e / e.sum(dim=-1, keepdim=True)
    return (pr @ v).transpose(1, 2).reshape(B, Tq, Wd)


def siglip_vision_forward(sd: Dict[str, torch.Tensor], images: torch.Tensor, *, patch: int, heads: int, act: str = "gelu",
                          taps: Optional[List[torch.Tensor]] = None, normalize: bool = True) -> torch.Tensor:
    """images fp32 [B,3,S,S] (already normalised with mean = std = 0.5) -> [B,W] fp32."""
    t = "visual.trunk."
    actf = gelu if act == "gelu" else gelu_tanh
    w = sd[t + "patch_embed.proj.weight"].to(torch.float32)
    B, _, S, _ = images.shape
    g = S // patch
    Wd = w.shape[0]
    pt = images.reshape(B, 3, g, patch, g, patch).permute(0, 2, 4, 1, 3, 5).reshape(B, g * g, 3 * patch * patch)
    x = pt @ w.reshape(Wd, -1).t() + sd[t + "patch_embed.proj.bias"] + sd[t + "pos_embed"].reshape(1, g * g, Wd)
    n_layers = 0
    while f"{t}blocks.{n_layers}.norm1.weight" in sd:
        n_layers += 1
    for i in range(n_layers):
        p = f"{t}blocks.{i}."
        h = layer_norm(x, sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-6)
        qkv = h @ sd[p + "attn.qkv.weight"].t() + sd[p + "attn.qkv.bias"]
        q, k, v = qkv.split(Wd, dim=-1)
        x = x + _attention(q, k, v, heads) @ sd[p + "attn.proj.weight"].t() + sd[p + "attn.proj.bias"]
        h = layer_norm(x, sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-6)
        h = actf(h @ sd[p + "mlp.fc1.weight"].t() + sd[p + "mlp.fc1.bias"])
        x = x + h @ sd[p + "mlp.fc2.weight"].t() + sd[p + "mlp.fc2.bias"]
        if taps is not None:
            taps.append(x.clone())
    x = layer_norm(x, sd[t + "norm.weight"], sd[t + "norm.bias"], 1e-6)
    a = t + "attn_pool."
    q = (sd[a + "latent"].reshape(1, 1, Wd) @ sd[a + "q.weight"].t() + sd[a + "q.bias"]).expand(B, 1, Wd)
    kv = x @ sd[a + "kv.weight"].t() + sd[a + "kv.bias"]
    k, v = kv.split(Wd, dim=-1)
    y = _attention(q, k, v, heads) @ sd[a + "proj.weight"].t() + sd[a + "proj.bias"]
    h = layer_norm(y, sd[a + "norm.weight"], sd[a + "norm.bias"], 1e-6)
    y = y + gelu(h @ sd[a + "mlp.fc1.weight"].t() + sd[a + "mlp.fc1.bias"]) @ sd[a + "mlp.fc2.weight"].t() + sd[a + "mlp.fc2.bias"]
    out = y[:, 0]
    if normalize:
        out = out / torch.linalg.norm(out, dim=-1, keepdim=True)  # mlfoundation_openclip.py:100
    return out


def siglip_text_forward(sd: Dict[str, torch.Tensor], tokens: torch.Tensor, *, heads: int, act: str = "gelu",
                        taps: Optional[List[torch.Tensor]] = None, normalize: bool = True) -> torch.Tensor:
    """tokens int [B,T] (right-padded with id 1, every position attended) -> [B,D] fp32."""
    tok = tokens.to(torch.int64)
    B, T = tok.shape
    actf = gelu if act == "gelu" else gelu_tanh
    x = sd["text.token_embedding.weight"].to(torch.float32)[tok] + sd["text.positional_embedding"].to(torch.float32)[:T]
    Wd = x.shape[-1]
    n_layers = 0
    while f"text.transformer.resblocks.{n_layers}.ln_1.weight" in sd:
        n_layers += 1
    for i in range(n_layers):
        p = f"text.transformer.resblocks.{i}."
        h = layer_norm(x, sd[p + "ln_1.weight"], sd[p + "ln_1.bias"], 1e-6)
        qkv = h @ sd[p + "attn.in_proj_weight"].t() + sd[p + "attn.in_proj_bias"]
        q, k, v = qkv.split(Wd, dim=-1)
        x = x + _attention(q, k, v, heads) @ sd[p + "attn.out_proj.weight"].t() + sd[p + "attn.out_proj.bias"]
        h = layer_norm(x, sd[p + "ln_2.weight"], sd[p + "ln_2.bias"], 1e-6)
        h = actf(h @ sd[p + "mlp.c_fc.weight"].t() + sd[p + "mlp.c_fc.bias"])
        x = x + h @ sd[p + "mlp.c_proj.weight"].t() + sd[p + "mlp.c_proj.bias"]
        if taps is not None:
            taps.append(x.clone())
    x = layer_norm(x, sd["text.ln_final.weight"], sd["text.ln_final.bias"], 1e-6)
    out = x[:, -1] @ sd["text.text_projection.weight"].t() + sd["text.text_projection.bias"]
    if normalize:
        out = out / torch.linalg.norm(out, dim=-1, keepdim=True)  # mlfoundation_openclip.py:107
    return out
