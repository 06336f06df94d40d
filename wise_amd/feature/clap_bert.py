"""Host side of MS-CLAP (version 2022)'s caption encoder: bert-base-uncased + msclap Projection, reached from the
reference at src/feature/microsoft_clap.py:53-58 when the feature id's version token is '2022'.

The encoder runs on the XLM-RoBERTa tower's kernels (`wise_xlmr_forward`) with the BERT switches of `wise_xlmr_config`
(absolute positions, LayerNorm eps 1e-12, [CLS] pooling, msclap Projection head).  Weights are addressed by msclap 1.3.3
state-dict keys under `clap.caption_encoder.` (`base.…` = transformers' BertModel, `projection.…`), so a real checkpoint
is a pure data problem: `pack_clap_bert_weights`.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict

import torch

from .. import _lib


@dataclass(frozen=True)
class ClapBertSpec:
    name: str = "clap-2022-bert-base-uncased"
    width: int = 768
    heads: int = 12
    layers: int = 12
    mlp: int = 3072
    embed_dim: int = 1024       # msclap d_proj
    vocab: int = 30522
    max_positions: int = 512
    context: int = 100          # msclap config_2022 text_len
    pad_id: int = 0             # [PAD]

    @property
    def proj_hidden(self) -> int:
        return self.embed_dim

    def flops_per_query(self) -> int:
        T, W, F = self.context, self.width, self.mlp
        per_layer = T * W * 3 * W * 2 + 2 * self.heads * T * T * 64 * 2 + T * W * W * 2 + 2 * T * W * F * 2
        return self.layers * per_layer + (W * self.embed_dim + self.embed_dim * self.embed_dim) * 2

    def c_config(self) -> _lib.XlmrConfig:
        # pos_mode 1 (absolute positions), pool 1 ([CLS]), head 1 (msclap Projection), eps 1e-12
        return _lib.XlmrConfig(self.context, self.vocab, self.max_positions, self.width, self.layers, self.heads, self.mlp,
                               self.proj_hidden, self.embed_dim, self.pad_id, 1, 1, 1, 1)


CLAP_BERT_SPEC = ClapBertSpec()


def clap_bert_state_dict_keys(spec: ClapBertSpec):
    """(key, shape) in the order the seeded initialiser draws them (msclap names; BERT's tanh pooler, which msclap's
    TextEncoder never reads, is not part of the list)."""
    W, F, V, P, D = spec.width, spec.mlp, spec.vocab, spec.max_positions, spec.embed_dim
    e = "base.embeddings."
    keys = [(e + "word_embeddings.weight", (V, W)), (e + "position_embeddings.weight", (P, W)),
            (e + "token_type_embeddings.weight", (2, W)), (e + "LayerNorm.weight", (W,)), (e + "LayerNorm.bias", (W,))]
    for i in range(spec.layers):
        p = f"base.encoder.layer.{i}."
        for n in ("query", "key", "value"):
            keys += [(p + f"attention.self.{n}.weight", (W, W)), (p + f"attention.self.{n}.bias", (W,))]
        keys += [(p + "attention.output.dense.weight", (W, W)), (p + "attention.output.dense.bias", (W,)),
                 (p + "attention.output.LayerNorm.weight", (W,)), (p + "attention.output.LayerNorm.bias", (W,)),
                 (p + "intermediate.dense.weight", (F, W)), (p + "intermediate.dense.bias", (F,)),
                 (p + "output.dense.weight", (W, F)), (p + "output.dense.bias", (W,)),
                 (p + "output.LayerNorm.weight", (W,)), (p + "output.LayerNorm.bias", (W,))]
    keys += [("projection.linear1.weight", (D, W)), ("projection.linear2.weight", (D, D)),
             ("projection.layer_norm.weight", (D,)), ("projection.layer_norm.bias", (D,))]
    return keys


def random_clap_bert_state_dict(spec: ClapBertSpec = CLAP_BERT_SPEC, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Seeded fp32 weights (no checkpoints exist offline)."""
    g = torch.Generator().manual_seed(5000 + seed)
    W = spec.width
    sd = {}
    for key, shape in clap_bert_state_dict_keys(spec):
        n = torch.randn(shape, generator=g, dtype=torch.float32)
        if key.endswith("word_embeddings.weight"):
            t = n * 0.5
        elif key.endswith("position_embeddings.weight") or key.endswith("token_type_embeddings.weight"):
            t = n * 0.25
        elif key.endswith("LayerNorm.weight") or key.endswith("layer_norm.weight"):
            t = 1.0 + 0.1 * n
        elif key.endswith("LayerNorm.bias") or key.endswith("layer_norm.bias"):
            t = 0.1 * n
        elif ".query.weight" in key or ".key.weight" in key:
            t = n * (W ** -0.5) * 2.0
        elif key.endswith(".weight"):
            t = n * (shape[1] ** -0.5)
        elif key.endswith(".bias"):
            t = 0.02 * n
        else:
            raise KeyError(key)
        sd[key] = t.contiguous()
    return sd


def pack_clap_bert_weights(spec: ClapBertSpec, sd: Dict[str, torch.Tensor]):
    """state dict -> (bf16 blob, fp32 blob) in wise_xlmr_forward's layout with head = 1 (include/wise_hip.h)."""
    f32 = lambda k: sd[k].detach().to(torch.float32).cpu()
    e = "base.embeddings."
    wb = []
    pf = [f32(e + "word_embeddings.weight").reshape(-1), f32(e + "position_embeddings.weight").reshape(-1),
          f32(e + "token_type_embeddings.weight")[0], f32(e + "LayerNorm.weight"), f32(e + "LayerNorm.bias")]
    for i in range(spec.layers):
        p = f"base.encoder.layer.{i}."
        wb += [torch.cat([f32(p + f"attention.self.{n}.weight") for n in ("query", "key", "value")]).reshape(-1),
               f32(p + "attention.output.dense.weight").reshape(-1), f32(p + "intermediate.dense.weight").reshape(-1),
               f32(p + "output.dense.weight").reshape(-1)]
        pf += [torch.cat([f32(p + f"attention.self.{n}.bias") for n in ("query", "key", "value")]),
               f32(p + "attention.output.dense.bias"), f32(p + "attention.output.LayerNorm.weight"),
               f32(p + "attention.output.LayerNorm.bias"), f32(p + "intermediate.dense.bias"), f32(p + "output.dense.bias"),
               f32(p + "output.LayerNorm.weight"), f32(p + "output.LayerNorm.bias")]
    wb += [f32("projection.linear1.weight").reshape(-1), f32("projection.linear2.weight").reshape(-1)]
    pf += [f32("projection.layer_norm.weight"), f32("projection.layer_norm.bias")]
    return torch.cat(wb).to(torch.bfloat16).contiguous(), torch.cat(pf).contiguous()
