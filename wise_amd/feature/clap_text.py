"""Host side of the MS-CLAP caption encoder: the text half of `microsoft/clap/2023`.

Replaces `self.model.clap.caption_encoder(preprocessed_text)` + L2 normalise at
src/feature/microsoft_clap.py:53-58.  In msclap 1.3.3 (un-vendored, requirements.txt:25-26) the 2023 caption
encoder is `TextEncoder`: a Hugging Face GPT-2 base (`AutoModel.from_pretrained('gpt2')`: 12 layers, width 768,
12 heads, gelu_new, context 77 of the 1024 learned positions), the hidden state of the last token that is not
the pad id 0, and msclap's `Projection` (768 -> 1024: linear1, GELU, linear2, LayerNorm(e1 + e2)).
The kernels are the CLIP text tower's with three switches (include/wise_hip.h: act = 2, pool = 1, head = 1).
Weights are addressed by msclap state-dict keys relative to `caption_encoder.` (`base.*` are GPT2Model's own
keys, Conv1D weights stored [in, out]).
"""
from __future__ import annotations

from typing import Dict

import torch

from .text import TextSpec

CAPTION_SPEC = TextSpec("clap-2023-gpt2", 768, 12, 12, 1024, context=77, vocab=50257, act="gelu_new",
                        pool="last_nonzero", head="clap")
GPT2_POSITIONS = 1024


def caption_state_dict_keys(spec: TextSpec = CAPTION_SPEC, positions: int = GPT2_POSITIONS):
    """(key, shape) in the order the seeded initialiser draws them."""
    W, F, D, V = spec.width, spec.mlp, spec.embed_dim, spec.vocab
    keys = [("base.wte.weight", (V, W)), ("base.wpe.weight", (positions, W))]
    for i in range(spec.layers):
        p = f"base.h.{i}."
        keys += [(p + "ln_1.weight", (W,)), (p + "ln_1.bias", (W,)), (p + "attn.c_attn.weight", (W, 3 * W)),
                 (p + "attn.c_attn.bias", (3 * W,)), (p + "attn.c_proj.weight", (W, W)), (p + "attn.c_proj.bias", (W,)),
                 (p + "ln_2.weight", (W,)), (p + "ln_2.bias", (W,)), (p + "mlp.c_fc.weight", (W, F)),
                 (p + "mlp.c_fc.bias", (F,)), (p + "mlp.c_proj.weight", (F, W)), (p + "mlp.c_proj.bias", (W,))]
    keys += [("base.ln_f.weight", (W,)), ("base.ln_f.bias", (W,)), ("projection.linear1.weight", (D, W)),
             ("projection.linear2.weight", (D, D)), ("projection.layer_norm.weight", (D,)),
             ("projection.layer_norm.bias", (D,))]
    return keys


def random_caption_state_dict(spec: TextSpec = CAPTION_SPEC, seed: int = 0,
                              positions: int = GPT2_POSITIONS) -> Dict[str, torch.Tensor]:
    """Seeded fp32 weights (no checkpoint exists offline); one CPU generator, `caption_state_dict_keys` order."""
    g = torch.Generator().manual_seed(2000 + seed)
    W, L = spec.width, max(spec.layers, 1)
    sd = {}
    for key, shape in caption_state_dict_keys(spec, positions):
        n = torch.randn(shape, generator=g, dtype=torch.float32)
        if key == "base.wte.weight":
            t = n * 0.5
        elif key == "base.wpe.weight":
            t = n * 0.25
        elif key.endswith("ln_1.weight") or key.endswith("ln_2.weight") or key.endswith("ln_f.weight") \
                or key.endswith("layer_norm.weight"):
            t = 1.0 + 0.1 * n
        elif ("ln_" in key or "layer_norm" in key) and key.endswith(".bias"):
            t = 0.1 * n
        elif key.endswith("c_attn.weight"):
            t = n * (W ** -0.5)
            t[:, : 2 * W] *= 2.0
        elif key.endswith("attn.c_proj.weight"):
            t = n * (W ** -0.5) * ((2 * L) ** -0.5)
        elif key.endswith("c_fc.weight"):
            t = n * (W ** -0.5)
        elif key.endswith("mlp.c_proj.weight"):
            t = n * (spec.mlp ** -0.5) * ((2 * L) ** -0.5)
        elif key.endswith(".bias"):
            t = 0.02 * n
        elif key == "projection.linear1.weight":
            t = n * (W ** -0.5)
        elif key == "projection.linear2.weight":
            t = n * (spec.embed_dim ** -0.5)
        else:
            raise KeyError(key)
        sd[key] = t.contiguous()
    return sd


def pack_caption_weights(spec: TextSpec, sd: Dict[str, torch.Tensor]):
    """msclap caption-encoder state dict -> (bf16 blob, fp32 blob) in the layout include/wise_hip.h documents.
    GPT-2's Conv1D weights are [in, out]; the kernels want [out, in] (K-contiguous), hence the transposes."""
    f32 = lambda k: sd[k].detach().to(torch.float32).cpu()
    lin = lambda k: f32(k).t().contiguous().reshape(-1)
    wb = []
    pf = [f32("base.wte.weight").reshape(-1), f32("base.wpe.weight")[: spec.context].reshape(-1)]
    for i in range(spec.layers):
        p = f"base.h.{i}."
        wb += [lin(p + "attn.c_attn.weight"), lin(p + "attn.c_proj.weight"), lin(p + "mlp.c_fc.weight"),
               lin(p + "mlp.c_proj.weight")]
        pf += [f32(p + "ln_1.weight"), f32(p + "ln_1.bias"), f32(p + "attn.c_attn.bias"), f32(p + "attn.c_proj.bias"),
               f32(p + "ln_2.weight"), f32(p + "ln_2.bias"), f32(p + "mlp.c_fc.bias"), f32(p + "mlp.c_proj.bias")]
    wb += [f32("projection.linear1.weight").reshape(-1), f32("projection.linear2.weight").reshape(-1)]
    pf += [f32("base.ln_f.weight"), f32("base.ln_f.bias"), f32("projection.layer_norm.weight"),
           f32("projection.layer_norm.bias")]
    return torch.cat(wb).to(torch.bfloat16).contiguous(), torch.cat(pf).contiguous()
