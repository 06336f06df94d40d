"""Host side of open_clip's SigLIP models (`ViT-*-SigLIP-*/webli`): the reference's own end-to-end test extracts video
features with `mlfoundations/open_clip/ViT-L-16-SigLIP-384/webli` (/root/reference/tests/test-kinetics-6.sh:91; its
factory docstring names `ViT-B-16-SigLIP-256`, src/feature/feature_extractor_factory.py:12).

Image tower: open_clip wraps a timm ViT (`TimmModel`, state-dict prefix `visual.trunk.`): patch embedding WITH bias, no
class token, no pre-norm, LayerNorm eps 1e-6, final norm over all tokens, attention-pool ('map') head
(timm AttentionPoolLatent: one latent query, q / kv / proj Linear layers, y + mlp(norm(y))), no projection.
Text tower: open_clip's own TextTransformer (prefix `text.`) without a causal mask, LayerNorm eps 1e-6, pooled at the
LAST position of the 64-token context, `text_projection` a Linear WITH bias; tokens from the model's sentencepiece
vocabulary (HF tokenizer, open_clip 'canonicalize' cleaning: lower case, punctuation removed).
GELU: timm 0.9.x (what open_clip 2.24.0 pulls in) builds these towers with nn.GELU (erf); the tanh form the checkpoints
were trained with is `act='gelu_tanh'` (both are kernels of this library) — UNPINNED offline which one a given
installation runs; the two differ by at most 5e-4 per activation.
"""
from __future__ import annotations

import os
import re
import string
from pathlib import Path
from typing import Dict, List, Union

import torch

from .text import TextSpec
from .vit import VitSpec

# (image tower, text tower) per open_clip model name
SIGLIP_VISION: Dict[str, VitSpec] = {
    "ViT-B-16-SigLIP": VitSpec("ViT-B-16-SigLIP", 224, 16, 768, 12, 12, 3072, 768, "gelu", 1),
    "ViT-B-16-SigLIP-256": VitSpec("ViT-B-16-SigLIP-256", 256, 16, 768, 12, 12, 3072, 768, "gelu", 1),
    "ViT-B-16-SigLIP-384": VitSpec("ViT-B-16-SigLIP-384", 384, 16, 768, 12, 12, 3072, 768, "gelu", 1),
    "ViT-L-16-SigLIP-256": VitSpec("ViT-L-16-SigLIP-256", 256, 16, 1024, 24, 16, 4096, 1024, "gelu", 1),
    "ViT-L-16-SigLIP-384": VitSpec("ViT-L-16-SigLIP-384", 384, 16, 1024, 24, 16, 4096, 1024, "gelu", 1),
}
SIGLIP_TEXT: Dict[str, TextSpec] = {
    name: TextSpec(name, v.width, v.heads, v.layers, v.embed_dim, context=64, vocab=32000, act="gelu", pool="last",
                   head="linear_bias", causal=False, ln_eps=1e-6)
    for name, v in SIGLIP_VISION.items()
}
SIGLIP_MEAN = (0.5, 0.5, 0.5)
SIGLIP_STD = (0.5, 0.5, 0.5)


def siglip_vision_keys(spec: VitSpec):
    """(key, shape) of the image tower in the order the seeded initialiser draws them (open_clip / timm names)."""
    W, F, T, P = spec.width, spec.mlp, spec.tokens, spec.patch
    t = "visual.trunk."
    keys = [(t + "patch_embed.proj.weight", (W, 3, P, P)), (t + "patch_embed.proj.bias", (W,)), (t + "pos_embed", (1, T, W))]
    for i in range(spec.layers):
        p = f"{t}blocks.{i}."
        keys += [(p + "norm1.weight", (W,)), (p + "norm1.bias", (W,)), (p + "attn.qkv.weight", (3 * W, W)),
                 (p + "attn.qkv.bias", (3 * W,)), (p + "attn.proj.weight", (W, W)), (p + "attn.proj.bias", (W,)),
                 (p + "norm2.weight", (W,)), (p + "norm2.bias", (W,)), (p + "mlp.fc1.weight", (F, W)),
                 (p + "mlp.fc1.bias", (F,)), (p + "mlp.fc2.weight", (W, F)), (p + "mlp.fc2.bias", (W,))]
    a = t + "attn_pool."
    keys += [(t + "norm.weight", (W,)), (t + "norm.bias", (W,)), (a + "latent", (1, 1, W)), (a + "q.weight", (W, W)),
             (a + "q.bias", (W,)), (a + "kv.weight", (2 * W, W)), (a + "kv.bias", (2 * W,)), (a + "proj.weight", (W, W)),
             (a + "proj.bias", (W,)), (a + "norm.weight", (W,)), (a + "norm.bias", (W,)), (a + "mlp.fc1.weight", (F, W)),
             (a + "mlp.fc1.bias", (F,)), (a + "mlp.fc2.weight", (W, F)), (a + "mlp.fc2.bias", (W,))]
    return keys


def siglip_text_keys(spec: TextSpec):
    W, F, D, T, V = spec.width, spec.mlp, spec.embed_dim, spec.context, spec.vocab
    keys = [("text.token_embedding.weight", (V, W)), ("text.positional_embedding", (T, W))]
    for i in range(spec.layers):
        p = f"text.transformer.resblocks.{i}."
        keys += [(p + "ln_1.weight", (W,)), (p + "ln_1.bias", (W,)), (p + "attn.in_proj_weight", (3 * W, W)),
                 (p + "attn.in_proj_bias", (3 * W,)), (p + "attn.out_proj.weight", (W, W)),
                 (p + "attn.out_proj.bias", (W,)), (p + "ln_2.weight", (W,)), (p + "ln_2.bias", (W,)),
                 (p + "mlp.c_fc.weight", (F, W)), (p + "mlp.c_fc.bias", (F,)), (p + "mlp.c_proj.weight", (W, F)),
                 (p + "mlp.c_proj.bias", (W,))]
    keys += [("text.ln_final.weight", (W,)), ("text.ln_final.bias", (W,)), ("text.text_projection.weight", (D, W)),
             ("text.text_projection.bias", (D,))]
    return keys


def _seeded(keys, seed: int, W: int, F: int, L: int, kdim: int = 0) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    L = max(L, 1)
    sd = {}
    for key, shape in keys:
        n = torch.randn(shape, generator=g, dtype=torch.float32)
        leaf = key.rsplit(".", 1)[-1]
        if key.endswith("patch_embed.proj.weight"):
            t = n * (kdim ** -0.5)
        elif key.endswith("pos_embed") or key.endswith("positional_embedding") or key.endswith("token_embedding.weight") \
                or key.endswith("latent"):
            t = n * 0.5
        elif leaf == "weight" and len(shape) == 1:                                 # LayerNorm scales
            t = 1.0 + 0.1 * n
        elif leaf == "bias" and any(s_ in key for s_ in ("norm", "ln_")):
            t = 0.1 * n
        elif key.endswith("attn.qkv.weight") or key.endswith("in_proj_weight"):
            t = n * (W ** -0.5)
            t[: 2 * W] *= 2.0
        elif key.endswith("attn_pool.q.weight") or key.endswith("attn_pool.kv.weight"):
            t = n * (W ** -0.5) * 1.5
        elif key.endswith("mlp.fc2.weight") or key.endswith("c_proj.weight"):
            t = n * (F ** -0.5) * ((2 * L) ** -0.5)
        elif key.endswith("attn.proj.weight") or key.endswith("out_proj.weight"):
            t = n * (W ** -0.5) * ((2 * L) ** -0.5)
        elif leaf == "weight":
            t = n * (shape[-1] ** -0.5)
        elif leaf in ("bias", "in_proj_bias"):
            t = 0.02 * n
        else:
            raise KeyError(key)
        sd[key] = t.contiguous()
    return sd


def random_siglip_vision_state_dict(spec: VitSpec, seed: int = 0) -> Dict[str, torch.Tensor]:
    return _seeded(siglip_vision_keys(spec), 5000 + seed, spec.width, spec.mlp, spec.layers, spec.kdim)


def random_siglip_text_state_dict(spec: TextSpec, seed: int = 0) -> Dict[str, torch.Tensor]:
    return _seeded(siglip_text_keys(spec), 6000 + seed, spec.width, spec.mlp, spec.layers)


def pack_siglip_vision(spec: VitSpec, sd: Dict[str, torch.Tensor]):
    """timm state dict -> (bf16 blob, fp32 blob) in the arch-1 layout of wise_vit_config (csrc/vit.hip vit_offsets):
    bf16 conv [W,Kp], per layer qkv / proj / fc1 / fc2, head kv [2W,W] / proj / fc1 / fc2;
    fp32 pos [T,W] (+ the patch embedding's bias), per layer (norm1 w,b, qkv b, proj b, norm2 w,b, fc1 b, fc2 b),
    final norm w,b, q(latent) [W] (the head's query is a constant of the weights), kv b, proj b, head norm w,b, fc1 b, fc2 b."""
    W = spec.width
    f32 = lambda k: sd[k].detach().to(torch.float32).cpu()
    t = "visual.trunk."
    conv = torch.zeros(W, spec.kpad, dtype=torch.float32)
    conv[:, : spec.kdim] = f32(t + "patch_embed.proj.weight").reshape(W, spec.kdim)
    wb = [conv.reshape(-1)]
    pf = [(f32(t + "pos_embed").reshape(spec.tokens, W) + f32(t + "patch_embed.proj.bias")[None, :]).reshape(-1)]
    for i in range(spec.layers):
        p = f"{t}blocks.{i}."
        wb += [f32(p + "attn.qkv.weight").reshape(-1), f32(p + "attn.proj.weight").reshape(-1),
               f32(p + "mlp.fc1.weight").reshape(-1), f32(p + "mlp.fc2.weight").reshape(-1)]
        pf += [f32(p + "norm1.weight"), f32(p + "norm1.bias"), f32(p + "attn.qkv.bias"), f32(p + "attn.proj.bias"),
               f32(p + "norm2.weight"), f32(p + "norm2.bias"), f32(p + "mlp.fc1.bias"), f32(p + "mlp.fc2.bias")]
    a = t + "attn_pool."
    qv = f32(a + "latent").reshape(W) @ f32(a + "q.weight").t() + f32(a + "q.bias")
    wb += [f32(a + "kv.weight").reshape(-1), f32(a + "proj.weight").reshape(-1), f32(a + "mlp.fc1.weight").reshape(-1),
           f32(a + "mlp.fc2.weight").reshape(-1)]
    pf += [f32(t + "norm.weight"), f32(t + "norm.bias"), qv, f32(a + "kv.bias"), f32(a + "proj.bias"), f32(a + "norm.weight"),
           f32(a + "norm.bias"), f32(a + "mlp.fc1.bias"), f32(a + "mlp.fc2.bias")]
    return torch.cat(wb).to(torch.bfloat16).contiguous(), torch.cat(pf).contiguous()


def pack_siglip_text(spec: TextSpec, sd: Dict[str, torch.Tensor]):
    """open_clip `text.` state dict -> the blobs of wise_text_forward (head = linear with bias: the bias follows ln_final)."""
    f32 = lambda k: sd[k].detach().to(torch.float32).cpu()
    wb = []
    pf = [f32("text.token_embedding.weight").reshape(-1), f32("text.positional_embedding").reshape(-1)]
    for i in range(spec.layers):
        p = f"text.transformer.resblocks.{i}."
        wb += [f32(p + "attn.in_proj_weight").reshape(-1), f32(p + "attn.out_proj.weight").reshape(-1),
               f32(p + "mlp.c_fc.weight").reshape(-1), f32(p + "mlp.c_proj.weight").reshape(-1)]
        pf += [f32(p + "ln_1.weight"), f32(p + "ln_1.bias"), f32(p + "attn.in_proj_bias"), f32(p + "attn.out_proj.bias"),
               f32(p + "ln_2.weight"), f32(p + "ln_2.bias"), f32(p + "mlp.c_fc.bias"), f32(p + "mlp.c_proj.bias")]
    wb.append(f32("text.text_projection.weight").reshape(-1))       # nn.Linear weight is already [D, W]
    pf += [f32("text.ln_final.weight"), f32("text.ln_final.bias"), f32("text.text_projection.bias")]
    return torch.cat(wb).to(torch.bfloat16).contiguous(), torch.cat(pf).contiguous()


def canonicalize(text: str) -> str:
    """open_clip tokenizer.py `_clean_canonicalize`: canonicalize_text(basic_clean(x)) — punctuation removed, lower case,
    whitespace collapsed (big_vision's text canonicalisation; basic_clean without ftfy is html.unescape twice + strip)."""
    import html

    text = html.unescape(html.unescape(text)).strip()
    text = text.replace("_", " ")
    text = text.translate(str.maketrans("", "", string.punctuation))
    text = text.lower()
    text = re.sub(r"\s+", " ", text)
    return text.strip()


class SiglipTokenizer:
    """open_clip's HFTokenizer('timm/ViT-B-16-SigLIP', clean='canonicalize') restated: canonicalize -> sentencepiece pieces
    (the T5-style c4-en 32k vocabulary: sentencepiece ids used as they are, </s> = 1 appended) -> truncated to `context`
    (the </s> survives) -> right-padded with id 1 (the model's pad id).  Needs the model's `spiece.model`."""

    EOS = PAD = 1

    def __init__(self, model_file: Union[str, Path], context: int = 64):
        import sentencepiece as spm

        self.sp = spm.SentencePieceProcessor(model_file=str(model_file))
        self.context = int(context)
        self.vocab_size = self.sp.get_piece_size()

    @classmethod
    def default(cls, context: int = 64) -> "SiglipTokenizer":
        root = os.environ.get("WISE_AMD_WEIGHTS_DIR", "")
        path = Path(root) / "siglip" / "spiece.model"
        if not root or not path.exists():
            raise FileNotFoundError("SigLIP's sentencepiece vocabulary not found: set WISE_AMD_WEIGHTS_DIR and place it at "
                                    "$WISE_AMD_WEIGHTS_DIR/siglip/spiece.model")
        return cls(path, context)

    def encode(self, text: str) -> List[int]:
        ids = list(self.sp.encode(canonicalize(text)))[: self.context - 1]
        return ids + [self.EOS]

    def __call__(self, texts: Union[str, List[str]]) -> torch.Tensor:
        if isinstance(texts, str):
            texts = [texts]
        out = torch.full((len(texts), self.context), self.PAD, dtype=torch.int64)
        for r, t in enumerate(texts):
            ids = self.encode(t)
            out[r, : len(ids)] = torch.tensor(ids, dtype=torch.int64)
        return out
