"""TEST INFRASTRUCTURE — golden vectors for the MS-CLAP caption encoder.

Run in the build container:  python -m oracle.make_golden_clap_text
Pins oracle/clap_text_ref.py's GPT-2 body against transformers' GPT2Model on the same seeded weights, then stores the
oracle's outputs for seeded token batches in tests/golden/clap_text*.npz.
"""
from __future__ import annotations

from pathlib import Path

import numpy as np
import torch

from oracle import clap_text_ref
from wise_amd.feature.clap_text import CAPTION_SPEC, random_caption_state_dict
from wise_amd.feature.text import TextSpec

GOLD = Path(__file__).resolve().parents[1] / "tests" / "golden"
TINY = TextSpec("caption-tiny", 128, 2, 2, 1024, context=77, vocab=600, act="gelu_new", pool="last_nonzero",
                head="clap")


def seeded_caption_tokens(n: int, context: int, seed: int, vocab: int) -> np.ndarray:
    """w1 .. wk <eot> 0 ...: what msclap's preprocess_text yields (text + ' <|endoftext|>', padded with id 0);
    the last row fills the whole context."""
    rng = np.random.default_rng(seed)
    out = np.zeros((n, context), dtype=np.int32)
    for i in range(n):
        k = context - 1 if i == n - 1 else int(rng.integers(1, 24))
        out[i, :k] = rng.integers(1, vocab - 1, k)
        out[i, k] = vocab - 1
    return out


def pin_against_hf(spec: TextSpec, sd, tokens: torch.Tensor, positions: int, tol: float):
    from transformers import GPT2Config, GPT2Model

    cfg = GPT2Config(vocab_size=spec.vocab, n_positions=positions, n_embd=spec.width, n_layer=spec.layers,
                     n_head=spec.heads, activation_function="gelu_new", resid_pdrop=0.0, embd_pdrop=0.0,
                     attn_pdrop=0.0, layer_norm_epsilon=1e-5)
    m = GPT2Model(cfg).eval()
    new = {k[len("base."):]: v for k, v in sd.items() if k.startswith("base.")}
    missing, unexpected = m.load_state_dict(new, strict=False)
    missing = [k for k in missing if not k.endswith(".attn.bias") and not k.endswith("masked_bias")]
    assert not missing and not unexpected, (missing, unexpected)
    with torch.no_grad():
        hf = m(input_ids=tokens.to(torch.int64)).last_hidden_state
        ours = clap_text_ref.gpt2_hidden(sd, tokens, spec.heads)
    d = (hf - ours).abs().max().item()
    print(f"  pin {spec.name}: |oracle - GPT2Model| hidden {d:.3e} (scale {ours.abs().max().item():.2f})")
    assert d <= tol * max(ours.abs().max().item(), 1.0), "oracle does not match transformers GPT2Model"


def golden(spec: TextSpec, seed: int, n: int, tok_seed: int, fname: str, positions: int):
    sd = random_caption_state_dict(spec, seed, positions)
    tokens = torch.from_numpy(seeded_caption_tokens(n, spec.context, tok_seed, spec.vocab))
    pin_against_hf(spec, sd, tokens, positions, 2e-5)
    with torch.no_grad():
        out = clap_text_ref.caption_forward(sd, tokens, spec.heads)
    np.savez_compressed(GOLD / fname, out=out.numpy(), tokens=tokens.numpy(),
                        meta=np.asarray([seed, n, tok_seed, positions], dtype=np.int64))
    print("wrote", fname, out.shape)


def main():
    golden(TINY, 0, 5, 51, "clap_text_tiny.npz", 96)
    golden(CAPTION_SPEC, 0, 4, 52, "clap_text.npz", 1024)


if __name__ == "__main__":
    main()
