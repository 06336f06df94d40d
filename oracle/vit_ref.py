"""ORACLE (test infrastructure — never imported by the product path under wise_amd/).

CPU fp32 restatement of the image path the reference runs at
/root/reference/src/feature/mlfoundation_openclip.py:92-101:

    model_output = self.model.encode_image(model_input).float()          (:99)
    model_output /= torch.linalg.norm(model_output, dim=-1, keepdims=True)  (:100)

`encode_image` lives in the un-vendored dependency open_clip_torch==2.24.0
(/root/reference/requirements.txt:11); its VisionTransformer.forward is restated here from the
published architecture: conv1 (stride=kernel=P, no bias) -> [cls | patches] +
positional_embedding -> ln_pre -> L x { x += out_proj(MHA(ln_1 x)); x += c_proj(act(c_fc(ln_2 x))) }
-> ln_post(cls) -> @ proj.

PINNING: open_clip itself is not installed and no checkpoint exists offline, so this oracle is
pinned against an independent implementation of the same architecture that IS in the container —
transformers' CLIPVisionModelWithProjection, fed the same seeded weights (oracle/make_golden.py,
max |diff| ~1e-6) — and against the shapes the reference's own test asserts
(src/feature/test_feature_extractor.py:33-34).  Parity with a real open_clip checkpoint is
UNPINNED offline.

Only plain tensor ops (matmul, exp, sum, erf) are used so that every step is explicit.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)  # open_clip OPENAI_DATASET_MEAN
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def layer_norm(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    mean = x.mean(dim=-1, keepdim=True)
    var = ((x - mean) ** 2).mean(dim=-1, keepdim=True)
    return (x - mean) / torch.sqrt(var + eps) * w + b


def quick_gelu(x: torch.Tensor) -> torch.Tensor:
    return x * torch.sigmoid(1.702 * x)


def gelu(x: torch.Tensor) -> torch.Tensor:
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def normalize_u8(frames_u8: torch.Tensor) -> torch.Tensor:
    """ToTensor + Normalize of the open_clip eval transform (resize/crop excluded): u8 [B,3,S,S] -> fp32."""
    x = frames_u8.to(torch.float32) / 255.0
    mean = torch.tensor(CLIP_MEAN, dtype=torch.float32).view(1, 3, 1, 1)
    std = torch.tensor(CLIP_STD, dtype=torch.float32).view(1, 3, 1, 1)
    return (x - mean) / std


def patchify(x: torch.Tensor, P: int) -> torch.Tensor:
    """[B,3,S,S] -> [B, g*g, 3*P*P] with k = c*P*P + py*P + px (the conv1 weight's flattening)."""
    B, Cc, S, _ = x.shape
    g = S // P
    x = x.reshape(B, Cc, g, P, g, P).permute(0, 2, 4, 1, 3, 5)  # B, gy, gx, c, py, px
    return x.reshape(B, g * g, Cc * P * P)


def vit_forward(sd: Dict[str, torch.Tensor], images: torch.Tensor, *, patch: int, heads: int, act: str = "quick_gelu",
                layers: Optional[int] = None, taps: Optional[List[torch.Tensor]] = None,
                normalize: bool = True) -> torch.Tensor:
    """images [B,3,S,S] fp32 (already normalised) -> [B,D] fp32.

    taps (if a list) receives the residual stream [B,T,W] after ln_pre and after every block.
    """
    x = images.to(torch.float32)
    B = x.shape[0]
    Wd = sd["visual.conv1.weight"].shape[0]
    conv_w = sd["visual.conv1.weight"].reshape(Wd, -1).to(torch.float32)
    x = patchify(x, patch) @ conv_w.t()  # [B, g*g, W]
    cls = sd["visual.class_embedding"].to(torch.float32).reshape(1, 1, Wd).expand(B, 1, Wd)
    x = torch.cat([cls, x], dim=1) + sd["visual.positional_embedding"].to(torch.float32)
    x = layer_norm(x, sd["visual.ln_pre.weight"], sd["visual.ln_pre.bias"])
    if taps is not None:
        taps.append(x.clone())
    T = x.shape[1]
    dh = Wd // heads
    n_layers = 0
    while f"visual.transformer.resblocks.{n_layers}.ln_1.weight" in sd:
        n_layers += 1
    if layers is not None:
        n_layers = min(n_layers, layers)
    actf = quick_gelu if act == "quick_gelu" else gelu
    for i in range(n_layers):
        p = f"visual.transformer.resblocks.{i}."
        h = layer_norm(x, sd[p + "ln_1.weight"], sd[p + "ln_1.bias"])
        qkv = h @ sd[p + "attn.in_proj_weight"].t() + sd[p + "attn.in_proj_bias"]
        q, k, v = qkv.split(Wd, dim=-1)
        q = q.reshape(B, T, heads, dh).transpose(1, 2)
        k = k.reshape(B, T, heads, dh).transpose(1, 2)
        v = v.reshape(B, T, heads, dh).transpose(1, 2)
        s = (q @ k.transpose(-1, -2)) / math.sqrt(dh)
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
    pooled = layer_norm(x[:, 0, :], sd["visual.ln_post.weight"], sd["visual.ln_post.bias"])
    out = pooled @ sd["visual.proj"].to(torch.float32)
    if normalize:
        out = out / torch.linalg.norm(out, dim=-1, keepdim=True)  # no epsilon: mlfoundation_openclip.py:100
    return out


def attention_ref(qkv: torch.Tensor, B: int, T: int, H: int, dh: int = 64) -> torch.Tensor:
    """qkv [B*T, 3*H*dh] -> o [B*T, H*dh]; the op wise_attention_bf16 / wise_attention_dh_bf16 computes."""
    Wd = H * dh
    q, k, v = qkv.to(torch.float32).reshape(B, T, 3 * Wd).split(Wd, dim=-1)
    q = q.reshape(B, T, H, dh).transpose(1, 2)
    k = k.reshape(B, T, H, dh).transpose(1, 2)
    v = v.reshape(B, T, H, dh).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) / math.sqrt(dh)
    pr = torch.softmax(s, dim=-1)
    return (pr @ v).transpose(1, 2).reshape(B * T, Wd)
