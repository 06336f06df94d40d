"""TEST INFRASTRUCTURE — golden vectors for the SigLIP towers (the model of the reference's end-to-end test).

Run in the build container:  python -m oracle.make_golden_siglip
Pins oracle/siglip_ref.py against transformers' SiglipVisionModel / SiglipTextModel on the same seeded weights, then stores
the oracle's outputs in tests/golden/siglip_*.npz.  Weights, frames and tokens are regenerated from seeds by the tests.
"""
from __future__ import annotations

from pathlib import Path

import numpy as np
import torch

from oracle import siglip_ref
from wise_amd.feature.siglip import (SIGLIP_TEXT, SIGLIP_VISION, random_siglip_text_state_dict,
                                     random_siglip_vision_state_dict)
from wise_amd.feature.text import TextSpec
from wise_amd.feature.vit import VitSpec

GOLD = Path(__file__).resolve().parents[1] / "tests" / "golden"
TINY_V = VitSpec("siglip-tiny", 64, 16, 128, 2, 2, 256, 128, "gelu", 1)              # 16 tokens
TINY_V_TANH = VitSpec("siglip-tiny-100", 160, 16, 128, 1, 2, 256, 128, "gelu_tanh", 1)   # 100 tokens: two key blocks
TINY_T = TextSpec("siglip-text-tiny", 128, 2, 2, 128, context=64, vocab=500, act="gelu", pool="last", head="linear_bias",
                  causal=False, ln_eps=1e-6)
HF_ACT = {"gelu": "gelu", "gelu_tanh": "gelu_pytorch_tanh"}


def normalize_u8(frames_u8: torch.Tensor) -> torch.Tensor:
    return (frames_u8.to(torch.float32) / 255.0 - 0.5) / 0.5


def seeded_frames(n: int, S: int, seed: int) -> np.ndarray:
    return np.random.default_rng(seed).integers(0, 256, size=(n, 3, S, S), dtype=np.uint8)


def seeded_tokens(n: int, spec: TextSpec, seed: int) -> np.ndarray:
    """w1 .. wk </s> <pad = 1> ...; the last row fills the context"""
    rng = np.random.default_rng(seed)
    out = np.ones((n, spec.context), dtype=np.int32)
    for i in range(n):
        k = spec.context - 1 if i == n - 1 else int(rng.integers(1, 20))
        out[i, :k] = rng.integers(2, spec.vocab, k)
        out[i, k] = 1
    return out


def pin_vision(spec: VitSpec, sd, x: torch.Tensor, tol: float):
    from transformers import SiglipVisionConfig, SiglipVisionModel

    cfg = SiglipVisionConfig(hidden_size=spec.width, intermediate_size=spec.mlp, num_hidden_layers=spec.layers,
                             num_attention_heads=spec.heads, image_size=spec.image_size, patch_size=spec.patch,
                             hidden_act=HF_ACT[spec.act], layer_norm_eps=1e-6, attention_dropout=0.0)
    m = SiglipVisionModel(cfg).eval()
    W = spec.width
    # (the wrapper's key prefix differs between transformers releases)
    t, v = "visual.trunk.", ("vision_model." if any(k.startswith("vision_model.") for k in m.state_dict()) else "")
    new = {v + "embeddings.patch_embedding.weight": sd[t + "patch_embed.proj.weight"],
           v + "embeddings.patch_embedding.bias": sd[t + "patch_embed.proj.bias"],
           v + "embeddings.position_embedding.weight": sd[t + "pos_embed"].reshape(-1, W),
           v + "post_layernorm.weight": sd[t + "norm.weight"], v + "post_layernorm.bias": sd[t + "norm.bias"]}
    for i in range(spec.layers):
        p, h = f"{t}blocks.{i}.", f"{v}encoder.layers.{i}."
        wq, wk, wv = sd[p + "attn.qkv.weight"].split(W, dim=0)
        bq, bk, bv = sd[p + "attn.qkv.bias"].split(W, dim=0)
        for n, w_, b_ in (("q", wq, bq), ("k", wk, bk), ("v", wv, bv)):
            new[h + f"self_attn.{n}_proj.weight"], new[h + f"self_attn.{n}_proj.bias"] = w_, b_
        new[h + "self_attn.out_proj.weight"], new[h + "self_attn.out_proj.bias"] = sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"]
        new[h + "layer_norm1.weight"], new[h + "layer_norm1.bias"] = sd[p + "norm1.weight"], sd[p + "norm1.bias"]
        new[h + "layer_norm2.weight"], new[h + "layer_norm2.bias"] = sd[p + "norm2.weight"], sd[p + "norm2.bias"]
        new[h + "mlp.fc1.weight"], new[h + "mlp.fc1.bias"] = sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"]
        new[h + "mlp.fc2.weight"], new[h + "mlp.fc2.bias"] = sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"]
    a, hd = t + "attn_pool.", v + "head."
    new[hd + "probe"] = sd[a + "latent"]
    new[hd + "attention.in_proj_weight"] = torch.cat([sd[a + "q.weight"], sd[a + "kv.weight"]])
    new[hd + "attention.in_proj_bias"] = torch.cat([sd[a + "q.bias"], sd[a + "kv.bias"]])
    new[hd + "attention.out_proj.weight"], new[hd + "attention.out_proj.bias"] = sd[a + "proj.weight"], sd[a + "proj.bias"]
    new[hd + "layernorm.weight"], new[hd + "layernorm.bias"] = sd[a + "norm.weight"], sd[a + "norm.bias"]
    new[hd + "mlp.fc1.weight"], new[hd + "mlp.fc1.bias"] = sd[a + "mlp.fc1.weight"], sd[a + "mlp.fc1.bias"]
    new[hd + "mlp.fc2.weight"], new[hd + "mlp.fc2.bias"] = sd[a + "mlp.fc2.weight"], sd[a + "mlp.fc2.bias"]
    missing, unexpected = m.load_state_dict(new, strict=False)
    assert not unexpected and not [k for k in missing if "position_ids" not in k], (missing, unexpected)
    with torch.no_grad():
        hf = m(pixel_values=x, output_hidden_states=True)
        taps = []
        out = siglip_ref.siglip_vision_forward(sd, x, patch=spec.patch, heads=spec.heads, act=spec.act, taps=taps,
                                               normalize=False)
    d_out = float((out - hf.pooler_output).abs().max())
    d_hid = max(float((a_ - b_).abs().max()) for a_, b_ in zip(taps, hf.hidden_states[1:]))
    scale = float(out.abs().max())
    print(f"  pin {spec.name}: |oracle - HF| out {d_out:.3e} (scale {scale:.2f}), hidden {d_hid:.3e}")
    # the HF head applies config.hidden_act in its MLP; timm's AttentionPoolLatent always erf GELU: equal only for act = gelu
    assert d_hid <= tol * 50 and (spec.act != "gelu" or d_out <= tol * max(scale, 1.0)), "oracle does not match transformers Siglip"
    return d_out, d_hid


def pin_text(spec: TextSpec, sd, tokens: torch.Tensor, tol: float):
    from transformers import SiglipTextConfig, SiglipTextModel

    cfg = SiglipTextConfig(vocab_size=spec.vocab, hidden_size=spec.width, intermediate_size=spec.mlp,
                           num_hidden_layers=spec.layers, num_attention_heads=spec.heads,
                           max_position_embeddings=spec.context, hidden_act=HF_ACT[spec.act], layer_norm_eps=1e-6,
                           attention_dropout=0.0, projection_size=spec.embed_dim)
    m = SiglipTextModel(cfg).eval()
    W = spec.width
    tm = "text_model." if any(k.startswith("text_model.") for k in m.state_dict()) else ""
    new = {tm + "embeddings.token_embedding.weight": sd["text.token_embedding.weight"],
           tm + "embeddings.position_embedding.weight": sd["text.positional_embedding"],
           tm + "final_layer_norm.weight": sd["text.ln_final.weight"],
           tm + "final_layer_norm.bias": sd["text.ln_final.bias"],
           tm + "head.weight": sd["text.text_projection.weight"], tm + "head.bias": sd["text.text_projection.bias"]}
    for i in range(spec.layers):
        p, h = f"text.transformer.resblocks.{i}.", f"{tm}encoder.layers.{i}."
        wq, wk, wv = sd[p + "attn.in_proj_weight"].split(W, dim=0)
        bq, bk, bv = sd[p + "attn.in_proj_bias"].split(W, dim=0)
        for n, w_, b_ in (("q", wq, bq), ("k", wk, bk), ("v", wv, bv)):
            new[h + f"self_attn.{n}_proj.weight"], new[h + f"self_attn.{n}_proj.bias"] = w_, b_
        new[h + "self_attn.out_proj.weight"], new[h + "self_attn.out_proj.bias"] = sd[p + "attn.out_proj.weight"], sd[p + "attn.out_proj.bias"]
        new[h + "layer_norm1.weight"], new[h + "layer_norm1.bias"] = sd[p + "ln_1.weight"], sd[p + "ln_1.bias"]
        new[h + "layer_norm2.weight"], new[h + "layer_norm2.bias"] = sd[p + "ln_2.weight"], sd[p + "ln_2.bias"]
        new[h + "mlp.fc1.weight"], new[h + "mlp.fc1.bias"] = sd[p + "mlp.c_fc.weight"], sd[p + "mlp.c_fc.bias"]
        new[h + "mlp.fc2.weight"], new[h + "mlp.fc2.bias"] = sd[p + "mlp.c_proj.weight"], sd[p + "mlp.c_proj.bias"]
    missing, unexpected = m.load_state_dict(new, strict=False)
    assert not unexpected and not [k for k in missing if "position_ids" not in k], (missing, unexpected)
    with torch.no_grad():
        hf = m(input_ids=tokens.long(), output_hidden_states=True)
        taps = []
        out = siglip_ref.siglip_text_forward(sd, tokens, heads=spec.heads, act=spec.act, taps=taps, normalize=False)
    d_out = float((out - hf.pooler_output).abs().max())
    d_hid = max(float((a_ - b_).abs().max()) for a_, b_ in zip(taps, hf.hidden_states[1:]))
    print(f"  pin {spec.name}: |oracle - HF| out {d_out:.3e}, hidden {d_hid:.3e}")
    assert d_out <= tol * max(float(out.abs().max()), 1.0) and d_hid <= tol * 50, "oracle does not match transformers Siglip text"
    return d_out, d_hid


def golden_vision(spec: VitSpec, seed: int, n: int, frame_seed: int, fname: str):
    print(f"[siglip vision] {spec.name}")
    sd = random_siglip_vision_state_dict(spec, seed)
    x = normalize_u8(torch.from_numpy(seeded_frames(n, spec.image_size, frame_seed)))
    torch.set_num_threads(8)
    pinned = pin_vision(spec, sd, x, 2e-5)
    taps = []
    with torch.no_grad():
        out = siglip_ref.siglip_vision_forward(sd, x, patch=spec.patch, heads=spec.heads, act=spec.act, taps=taps)
    taps_np = np.stack([t[:, 0, :].numpy() for t in taps])          # first token of every frame after every block
    np.savez_compressed(GOLD / fname, out=out.numpy(), taps=taps_np, meta=np.array([seed, n, frame_seed]),
                        pin_out=pinned[0], pin_hidden=pinned[1])
    print(f"  wrote {fname}: out {tuple(out.shape)}, taps {taps_np.shape}")


def golden_text(spec: TextSpec, seed: int, n: int, tok_seed: int, fname: str):
    print(f"[siglip text] {spec.name}")
    sd = random_siglip_text_state_dict(spec, seed)
    tokens = torch.from_numpy(seeded_tokens(n, spec, tok_seed))
    torch.set_num_threads(8)
    pinned = pin_text(spec, sd, tokens, 2e-5)
    taps = []
    with torch.no_grad():
        out = siglip_ref.siglip_text_forward(sd, tokens, heads=spec.heads, act=spec.act, taps=taps)
    taps_np = np.stack([t[:, -1, :].numpy() for t in taps])         # the pooled (last) position after every block
    np.savez_compressed(GOLD / fname, out=out.numpy(), taps=taps_np, tokens=tokens.numpy(), meta=np.array([seed, n, tok_seed]),
                        pin_out=pinned[0], pin_hidden=pinned[1])
    print(f"  wrote {fname}: out {tuple(out.shape)}, taps {taps_np.shape}")


def main():
    GOLD.mkdir(parents=True, exist_ok=True)
    golden_vision(TINY_V, 3, 3, 31, "siglip_v_tiny.npz")
    golden_vision(TINY_V_TANH, 4, 2, 32, "siglip_v_tiny100.npz")
    golden_text(TINY_T, 3, 5, 33, "siglip_t_tiny.npz")
    golden_vision(SIGLIP_VISION["ViT-B-16-SigLIP-256"], 0, 2, 36, "siglip_v_b16_256.npz")     # the id the factory docstring names
    golden_vision(SIGLIP_VISION["ViT-L-16-SigLIP-384"], 0, 2, 34, "siglip_v_l16_384.npz")
    golden_text(SIGLIP_TEXT["ViT-L-16-SigLIP-384"], 0, 3, 35, "siglip_t_l16_384.npz")


if __name__ == "__main__":
    main()
