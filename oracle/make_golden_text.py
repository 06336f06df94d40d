"""TEST INFRASTRUCTURE — golden vectors for the text tower.

Run in the build container:  python -m oracle.make_golden_text
Pins oracle/text_ref.py against transformers' CLIPTextModelWithProjection on the same seeded weights, then stores
the oracle's outputs for seeded token batches in tests/golden/text_*.npz.  Weights and tokens are regenerated
from seeds by the tests (wise_amd.feature.text.random_text_state_dict, seeded_tokens below).
"""
from __future__ import annotations

from pathlib import Path

import numpy as np
import torch

from oracle import text_ref
from wise_amd.feature.text import EOT_TOKEN, SOT_TOKEN, TextSpec, random_text_state_dict, text_spec_for

GOLD = Path(__file__).resolve().parents[1] / "tests" / "golden"


def seeded_tokens(n: int, context: int, seed: int, vocab: int = 49408) -> np.ndarray:
    """<sot> w1 .. wk <eot> 0 ...: lengths from 1 word to a full context (the last row is truncated-full)."""
    rng = np.random.default_rng(seed)
    out = np.zeros((n, context), dtype=np.int32)
    for i in range(n):
        k = context - 2 if i == n - 1 else int(rng.integers(1, min(24, context - 2) + 1))
        out[i, 0] = SOT_TOKEN if vocab > SOT_TOKEN else vocab - 2
        out[i, 1:1 + k] = rng.integers(1, min(vocab, SOT_TOKEN) - 1, k)
        out[i, 1 + k] = EOT_TOKEN if vocab > EOT_TOKEN else vocab - 1
    return out


def hf_model(spec: TextSpec, sd):
    from transformers import CLIPTextConfig, CLIPTextModelWithProjection

    cfg = CLIPTextConfig(vocab_size=spec.vocab, hidden_size=spec.width, intermediate_size=spec.mlp,
                         num_hidden_layers=spec.layers, num_attention_heads=spec.heads,
                         max_position_embeddings=spec.context, projection_dim=spec.embed_dim, hidden_act=spec.act,
                         attention_dropout=0.0, layer_norm_eps=1e-5, eos_token_id=2, bos_token_id=0, pad_token_id=1)
    m = CLIPTextModelWithProjection(cfg).eval()
    W = spec.width
    new = {"text_model.embeddings.token_embedding.weight": sd["token_embedding.weight"],
           "text_model.embeddings.position_embedding.weight": sd["positional_embedding"]}
    for i in range(spec.layers):
        p = f"transformer.resblocks.{i}."
        h = f"text_model.encoder.layers.{i}."
        wq, wk, wv = sd[p + "attn.in_proj_weight"].split(W, dim=0)
        bq, bk, bv = sd[p + "attn.in_proj_bias"].split(W, dim=0)
        for n, w_, b_ in (("q", wq, bq), ("k", wk, bk), ("v", wv, bv)):
            new[h + f"self_attn.{n}_proj.weight"] = w_
            new[h + f"self_attn.{n}_proj.bias"] = b_
        new[h + "self_attn.out_proj.weight"] = sd[p + "attn.out_proj.weight"]
        new[h + "self_attn.out_proj.bias"] = sd[p + "attn.out_proj.bias"]
        new[h + "layer_norm1.weight"] = sd[p + "ln_1.weight"]
        new[h + "layer_norm1.bias"] = sd[p + "ln_1.bias"]
        new[h + "layer_norm2.weight"] = sd[p + "ln_2.weight"]
        new[h + "layer_norm2.bias"] = sd[p + "ln_2.bias"]
        new[h + "mlp.fc1.weight"] = sd[p + "mlp.c_fc.weight"]
        new[h + "mlp.fc1.bias"] = sd[p + "mlp.c_fc.bias"]
        new[h + "mlp.fc2.weight"] = sd[p + "mlp.c_proj.weight"]
        new[h + "mlp.fc2.bias"] = sd[p + "mlp.c_proj.bias"]
    new["text_model.final_layer_norm.weight"] = sd["ln_final.weight"]
    new["text_model.final_layer_norm.bias"] = sd["ln_final.bias"]
    new["text_projection.weight"] = sd["text_projection"].t().contiguous()
    missing, unexpected = m.load_state_dict(new, strict=False)
    missing = [k for k in missing if "position_ids" not in k]
    assert not missing and not unexpected, (missing, unexpected)
    return m


def pin_against_hf(spec: TextSpec, sd, tokens: torch.Tensor, tol: float):
    m = hf_model(spec, sd)
    with torch.no_grad():
        hf = m(input_ids=tokens.to(torch.int64), output_hidden_states=True)
        taps = []
        ours = text_ref.text_forward(sd, tokens, heads=spec.heads, act=spec.act, taps=taps, normalize=False)
    d_out = (hf.text_embeds - ours).abs().max().item()
    d_hid = max((hf.hidden_states[i + 1] - taps[i]).abs().max().item() for i in range(spec.layers)) if spec.layers else 0
    scale = ours.abs().max().item()
    print(f"  pin {spec.name}: |oracle - HF| out {d_out:.3e} (scale {scale:.2f}), hidden {d_hid:.3e}")
    assert d_out <= tol * max(scale, 1.0) and d_hid <= tol * 50, "oracle does not match transformers CLIP text model"


def golden(spec: TextSpec, seed: int, n: int, tok_seed: int, fname: str):
    sd = random_text_state_dict(spec, seed)
    tokens = torch.from_numpy(seeded_tokens(n, spec.context, tok_seed, spec.vocab))
    pin_against_hf(spec, sd, tokens, 2e-5)
    with torch.no_grad():
        taps = []
        out = text_ref.text_forward(sd, tokens, heads=spec.heads, act=spec.act, taps=taps)
        raw = text_ref.text_forward(sd, tokens, heads=spec.heads, act=spec.act, normalize=False)
    np.savez_compressed(GOLD / fname, out=out.numpy(), raw=raw.numpy(), tokens=tokens.numpy(),
                        resid_last=taps[-1][:, :8, :].numpy() if taps else np.zeros(0, np.float32),
                        meta=np.asarray([seed, n, tok_seed, spec.width, spec.heads, spec.layers, spec.embed_dim,
                                         spec.context, spec.vocab], dtype=np.int64))
    print("wrote", fname, out.shape)


TINY = TextSpec("text-tiny", 128, 2, 2, 64, context=77, vocab=1000)
TINY_GELU = TextSpec("text-tiny-gelu", 256, 4, 3, 128, context=77, vocab=1000, act="gelu")


def main():
    golden(TINY, 0, 6, 31, "text_tiny.npz")
    golden(TINY_GELU, 1, 5, 32, "text_tiny_gelu.npz")
    golden(text_spec_for("ViT-B-32", "openai"), 0, 6, 33, "text_b32.npz")
    golden(text_spec_for("ViT-L-14", "openai"), 0, 3, 34, "text_l14.npz")
    golden(text_spec_for("ViT-H-14", "laion2b_s32b_b79k"), 0, 3, 35, "text_h14.npz")     # 24 layers x 1024, erf GELU


if __name__ == "__main__":
    main()
