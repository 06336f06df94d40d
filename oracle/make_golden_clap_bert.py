"""TEST INFRASTRUCTURE — golden vectors for MS-CLAP 2022's caption encoder (bert-base-uncased + msclap Projection).

Run in the build container:  python -m oracle.make_golden_clap_bert
Pins oracle/clap_bert_ref.py's encoder against transformers' BertModel on the same seeded weights (hidden states of the
live rows), then stores the oracle's outputs for seeded token batches in tests/golden/clap_bert_*.npz.  Weights and
tokens are regenerated from seeds by the tests.
"""
from __future__ import annotations

from pathlib import Path

import numpy as np
import torch

from oracle import clap_bert_ref
from wise_amd.feature.clap_bert import CLAP_BERT_SPEC, ClapBertSpec, random_clap_bert_state_dict

GOLD = Path(__file__).resolve().parents[1] / "tests" / "golden"
TINY = ClapBertSpec("clap-bert-tiny", 256, 4, 2, 512, 1024, vocab=1536, max_positions=128, context=100)


def seeded_tokens(n: int, spec: ClapBertSpec, seed: int) -> np.ndarray:
    """[CLS] w1 .. wk [SEP] [PAD] ...: lengths from 1 word to a full context (the last row is full)."""
    rng = np.random.default_rng(seed)
    out = np.full((n, spec.context), spec.pad_id, dtype=np.int32)
    for i in range(n):
        k = spec.context - 2 if i == n - 1 else int(rng.integers(1, 25))
        out[i, 0] = 101
        out[i, 1:1 + k] = rng.integers(104, spec.vocab, k)
        out[i, 1 + k] = 102
    return out


def hf_model(spec: ClapBertSpec, sd):
    from transformers import BertConfig, BertModel

    cfg = BertConfig(vocab_size=spec.vocab, hidden_size=spec.width, num_hidden_layers=spec.layers,
                     num_attention_heads=spec.heads, intermediate_size=spec.mlp, hidden_act="gelu",
                     hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0,
                     max_position_embeddings=spec.max_positions, type_vocab_size=2, layer_norm_eps=1e-12,
                     pad_token_id=spec.pad_id)
    m = BertModel(cfg, add_pooling_layer=False).eval()
    new = {k[len("base."):]: v for k, v in sd.items() if k.startswith("base.")}
    missing, unexpected = m.load_state_dict(new, strict=False)
    assert not unexpected and all("position_ids" in k or "token_type_ids" in k for k in missing), (missing, unexpected)
    return m


def pin_against_hf(spec: ClapBertSpec, sd, tokens: torch.Tensor, tol: float):
    m = hf_model(spec, sd)
    mask = (tokens != spec.pad_id).long()
    with torch.no_grad():
        hf = m(input_ids=tokens.long(), attention_mask=mask, token_type_ids=torch.zeros_like(tokens).long(),
               output_hidden_states=True)
        taps = []
        clap_bert_ref.bert_hidden(sd, tokens, heads=spec.heads, pad_id=spec.pad_id, taps=taps)
    live = mask.bool()
    d_hid = max(float((a[live] - b[live]).abs().max()) for a, b in zip(taps, hf.hidden_states))
    d_cls = float((taps[-1][:, 0] - hf.last_hidden_state[:, 0]).abs().max())
    print(f"  pin {spec.name}: |oracle - HF| CLS row {d_cls:.3e}, live hidden rows {d_hid:.3e}")
    assert d_cls <= tol and d_hid <= tol * 5, "oracle does not match transformers BertModel"
    return d_cls, d_hid


def golden(spec: ClapBertSpec, seed: int, n: int, tok_seed: int, fname: str):
    print(f"[clap-bert] {spec.name}")
    sd = random_clap_bert_state_dict(spec, seed)
    tokens = torch.from_numpy(seeded_tokens(n, spec, tok_seed))
    torch.set_num_threads(8)
    pinned = pin_against_hf(spec, sd, tokens, 5e-5)
    taps = []
    with torch.no_grad():
        clap_bert_ref.bert_hidden(sd, tokens, heads=spec.heads, pad_id=spec.pad_id, taps=taps)
        out = clap_bert_ref.caption_forward_2022(sd, tokens, heads=spec.heads)
    taps_np = np.stack([t[:, 0, :].numpy() for t in taps])     # the [CLS] row after the embeddings and every layer
    np.savez_compressed(GOLD / fname, out=out.numpy(), taps=taps_np, tokens=tokens.numpy(),
                        meta=np.array([seed, n, tok_seed]), pin_cls=pinned[0], pin_hidden=pinned[1])
    print(f"  wrote {fname}: out {tuple(out.shape)}, taps {taps_np.shape}, cos(out0,out1) {float((out[0] * out[1]).sum()):.3f}")


def main():
    GOLD.mkdir(parents=True, exist_ok=True)
    golden(TINY, 7, 6, 31, "clap_bert_tiny.npz")
    golden(CLAP_BERT_SPEC, 0, 3, 32, "clap_bert_base.npz")


if __name__ == "__main__":
    main()
