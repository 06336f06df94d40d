"""TEST INFRASTRUCTURE — golden vectors for the XLM-RoBERTa text tower (the reference's default model pair).

Run in the build container:  python -m oracle.make_golden_xlmr
Pins oracle/xlmr_text_ref.py against transformers' XLMRobertaModel on the same seeded weights (hidden states and the
mean-pooled, projected output computed from transformers' last_hidden_state), then stores the oracle's outputs for seeded
token batches in tests/golden/xlmr_*.npz.  Weights and tokens are regenerated from seeds by the tests.
"""
from __future__ import annotations

from pathlib import Path

import numpy as np
import torch

from oracle import xlmr_text_ref
from wise_amd.feature.xlmr_text import XLMR_SPECS, XlmrSpec, random_xlmr_state_dict

GOLD = Path(__file__).resolve().parents[1] / "tests" / "golden"
TINY = XlmrSpec("xlmr-tiny", 256, 4, 2, 512, 128, vocab=1000, max_positions=80, context=77)
TINY_SHORT = XlmrSpec("xlmr-tiny-short", 256, 4, 1, 256, 64, vocab=300, max_positions=40, context=32)


def seeded_tokens(n: int, spec: XlmrSpec, seed: int) -> np.ndarray:
    """<s> w1 .. wk </s> <pad> ...: lengths from 1 word to a full context (the last row is truncated-full)."""
    rng = np.random.default_rng(seed)
    out = np.full((n, spec.context), spec.pad_id, dtype=np.int32)
    for i in range(n):
        k = spec.context - 2 if i == n - 1 else int(rng.integers(1, min(24, spec.context - 2) + 1))
        out[i, 0] = 0
        out[i, 1:1 + k] = rng.integers(4, spec.vocab, k)
        out[i, 1 + k] = 2
    return out


def hf_model(spec: XlmrSpec, sd):
    from transformers import XLMRobertaConfig, XLMRobertaModel

    cfg = XLMRobertaConfig(vocab_size=spec.vocab, hidden_size=spec.width, num_hidden_layers=spec.layers,
                           num_attention_heads=spec.heads, intermediate_size=spec.mlp, hidden_act="gelu",
                           hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0,
                           max_position_embeddings=spec.max_positions, type_vocab_size=1, layer_norm_eps=1e-5,
                           pad_token_id=spec.pad_id, bos_token_id=0, eos_token_id=2)
    m = XLMRobertaModel(cfg, add_pooling_layer=False).eval()
    new = {k[len("text.transformer."):]: v for k, v in sd.items() if k.startswith("text.transformer.")}
    missing, unexpected = m.load_state_dict(new, strict=False)
    assert not unexpected and all("position_ids" in k or "token_type_ids" in k for k in missing), (missing, unexpected)
    return m


def pin_against_hf(spec: XlmrSpec, sd, tokens: torch.Tensor, tol: float):
    m = hf_model(spec, sd)
    mask = (tokens != spec.pad_id).long()
    with torch.no_grad():
        hf = m(input_ids=tokens.long(), attention_mask=mask, output_hidden_states=True)
        taps = []
        out = xlmr_text_ref.xlmr_text_forward(sd, tokens, heads=spec.heads, pad_id=spec.pad_id, taps=taps)
        # open_clip's MeanPooler and 'mlp' projection on transformers' own last_hidden_state
        mo = (hf.last_hidden_state * mask.unsqueeze(-1)).sum(dim=1) / mask.sum(-1, keepdim=True)
        pr = torch.nn.functional.gelu(mo @ sd["text.proj.0.weight"].t()) @ sd["text.proj.2.weight"].t()
        pr = pr / pr.norm(dim=-1, keepdim=True)
    live = mask.bool()
    d_hid = max(float((a[live] - b[live]).abs().max()) for a, b in zip(taps, hf.hidden_states))   # padded rows are don't-cares
    d_out = float((out - pr).abs().max())
    print(f"  pin {spec.name}: |oracle - HF| out {d_out:.3e}, hidden {d_hid:.3e}")
    assert d_out <= tol and d_hid <= tol * 50, "oracle does not match transformers XLMRobertaModel"
    return d_out, d_hid


def golden(spec: XlmrSpec, seed: int, n: int, tok_seed: int, fname: str):
    print(f"[xlmr] {spec.name}")
    sd = random_xlmr_state_dict(spec, seed)
    tokens = torch.from_numpy(seeded_tokens(n, spec, tok_seed))
    torch.set_num_threads(8)
    pinned = pin_against_hf(spec, sd, tokens, 2e-5)
    taps = []
    with torch.no_grad():
        out = xlmr_text_ref.xlmr_text_forward(sd, tokens, heads=spec.heads, pad_id=spec.pad_id, taps=taps)
    # taps: the first (<s>) row of every sequence after the embeddings and every layer
    taps_np = np.stack([t[:, 0, :].numpy() for t in taps])
    np.savez_compressed(GOLD / fname, out=out.numpy(), taps=taps_np, tokens=tokens.numpy(),
                        meta=np.array([seed, n, tok_seed]), pin_out=pinned[0], pin_hidden=pinned[1])
    print(f"  wrote {fname}: out {tuple(out.shape)}, taps {taps_np.shape}")


def main():
    GOLD.mkdir(parents=True, exist_ok=True)
    golden(TINY, 5, 6, 21, "xlmr_tiny.npz")
    golden(TINY_SHORT, 6, 4, 22, "xlmr_tiny_short.npz")
    golden(XLMR_SPECS["xlm-roberta-large-ViT-H-14"], 0, 3, 23, "xlmr_large.npz")
    golden(XLMR_SPECS["xlm-roberta-base-ViT-B-32"], 0, 3, 24, "xlmr_base.npz")


if __name__ == "__main__":
    main()
