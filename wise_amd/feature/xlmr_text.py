"""Host side of open_clip text towers that wrap a Hugging Face encoder: XLM-RoBERTa (`HFTextEncoder` with the mean
pooler and the two-layer MLP projection) — the text tower of `xlm-roberta-large-ViT-H-14`, the reference's default
feature id (/root/reference/extract-features.py:192), reached from src/feature/mlfoundation_openclip.py:103-108.

Weights are addressed by open_clip 2.24.0 state-dict keys (`text.transformer.…` = transformers' XLMRobertaModel,
`text.proj.0/2.weight` = the MLP projection), so a real checkpoint is a pure data problem: `pack_xlmr_weights`.
Tokenising is `XlmrTokenizer` (sentencepiece model file + the fairseq id shift Hugging Face applies).
"""
from __future__ import annotations

import ctypes as C
import html
import os
import re
from dataclasses import dataclass
from pathlib import Path
from typing import Dict, List, Union

import torch

from .. import _lib


@dataclass(frozen=True)
class XlmrSpec:
    name: str
    width: int
    heads: int
    layers: int
    mlp: int
    embed_dim: int
    vocab: int = 250002
    max_positions: int = 514
    context: int = 77          # open_clip tokenises to its own context_length, not the encoder's 512
    pad_id: int = 1

    @property
    def proj_hidden(self) -> int:
        return (self.width + self.embed_dim) // 2   # open_clip hf_model.py, proj_type 'mlp'

    def flops_per_query(self) -> int:
        T, W, F = self.context, self.width, self.mlp
        per_layer = T * W * 3 * W * 2 + 2 * self.heads * T * T * 64 * 2 + T * W * W * 2 + 2 * T * W * F * 2
        return self.layers * per_layer + (W * self.proj_hidden + self.proj_hidden * self.embed_dim) * 2

    def c_config(self) -> _lib.XlmrConfig:
        return _lib.XlmrConfig(self.context, self.vocab, self.max_positions, self.width, self.layers, self.heads, self.mlp,
                               self.proj_hidden, self.embed_dim, self.pad_id, 0, 0, 0, 0)


# open_clip model configs whose text_cfg names a Hugging Face XLM-RoBERTa (model_configs/xlm-roberta-*-ViT-*.json)
XLMR_SPECS: Dict[str, XlmrSpec] = {
    "xlm-roberta-large-ViT-H-14": XlmrSpec("xlm-roberta-large-ViT-H-14", 1024, 16, 24, 4096, 1024),
    "xlm-roberta-base-ViT-B-32": XlmrSpec("xlm-roberta-base-ViT-B-32", 768, 12, 12, 3072, 512),
}


def xlmr_state_dict_keys(spec: XlmrSpec):
    """(key, shape) in the order the seeded initialiser draws them (open_clip names)."""
    W, F, V, P = spec.width, spec.mlp, spec.vocab, spec.max_positions
    e = "text.transformer.embeddings."
    keys = [(e + "word_embeddings.weight", (V, W)), (e + "position_embeddings.weight", (P, W)),
            (e + "token_type_embeddings.weight", (1, W)), (e + "LayerNorm.weight", (W,)), (e + "LayerNorm.bias", (W,))]
    for i in range(spec.layers):
        p = f"text.transformer.encoder.layer.{i}."
        for n in ("query", "key", "value"):
            keys += [(p + f"attention.self.{n}.weight", (W, W)), (p + f"attention.self.{n}.bias", (W,))]
        keys += [(p + "attention.output.dense.weight", (W, W)), (p + "attention.output.dense.bias", (W,)),
                 (p + "attention.output.LayerNorm.weight", (W,)), (p + "attention.output.LayerNorm.bias", (W,)),
                 (p + "intermediate.dense.weight", (F, W)), (p + "intermediate.dense.bias", (F,)),
                 (p + "output.dense.weight", (W, F)), (p + "output.dense.bias", (W,)),
                 (p + "output.LayerNorm.weight", (W,)), (p + "output.LayerNorm.bias", (W,))]
    keys += [("text.proj.0.weight", (spec.proj_hidden, W)), ("text.proj.2.weight", (spec.embed_dim, spec.proj_hidden))]
    return keys


def random_xlmr_state_dict(spec: XlmrSpec, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Seeded fp32 weights (no checkpoints exist offline)."""
    g = torch.Generator().manual_seed(3000 + seed)
    W = spec.width
    sd = {}
    for key, shape in xlmr_state_dict_keys(spec):
        n = torch.randn(shape, generator=g, dtype=torch.float32)
        if key.endswith("word_embeddings.weight"):
            t = n * 0.5
        elif key.endswith("position_embeddings.weight") or key.endswith("token_type_embeddings.weight"):
            t = n * 0.25
        elif key.endswith("LayerNorm.weight"):
            t = 1.0 + 0.1 * n
        elif key.endswith("LayerNorm.bias"):
            t = 0.1 * n
        elif ".query.weight" in key or ".key.weight" in key:
            t = n * (W ** -0.5) * 2.0
        elif key.endswith("output.dense.weight") and "attention" not in key:
            t = n * (spec.mlp ** -0.5)
        elif key.endswith(".weight"):
            t = n * (shape[1] ** -0.5)
        elif key.endswith(".bias"):
            t = 0.02 * n
        else:
            raise KeyError(key)
        sd[key] = t.contiguous()
    return sd


def pack_xlmr_weights(spec: XlmrSpec, sd: Dict[str, torch.Tensor]):
    """state dict -> (bf16 blob, fp32 blob) in the layout include/wise_hip.h documents (CPU tensors)."""
    f32 = lambda k: sd[k].detach().to(torch.float32).cpu()
    e = "text.transformer.embeddings."
    wb = []
    pf = [f32(e + "word_embeddings.weight").reshape(-1), f32(e + "position_embeddings.weight").reshape(-1),
          f32(e + "token_type_embeddings.weight")[0], f32(e + "LayerNorm.weight"), f32(e + "LayerNorm.bias")]
    for i in range(spec.layers):
        p = f"text.transformer.encoder.layer.{i}."
        wb += [torch.cat([f32(p + f"attention.self.{n}.weight") for n in ("query", "key", "value")]).reshape(-1),
               f32(p + "attention.output.dense.weight").reshape(-1), f32(p + "intermediate.dense.weight").reshape(-1),
               f32(p + "output.dense.weight").reshape(-1)]
        pf += [torch.cat([f32(p + f"attention.self.{n}.bias") for n in ("query", "key", "value")]),
               f32(p + "attention.output.dense.bias"), f32(p + "attention.output.LayerNorm.weight"),
               f32(p + "attention.output.LayerNorm.bias"), f32(p + "intermediate.dense.bias"), f32(p + "output.dense.bias"),
               f32(p + "output.LayerNorm.weight"), f32(p + "output.LayerNorm.bias")]
    wb += [f32("text.proj.0.weight").reshape(-1), f32("text.proj.2.weight").reshape(-1)]
    return torch.cat(wb).to(torch.bfloat16).contiguous(), torch.cat(pf).contiguous()


class XlmrTextEngine:
    """Device copies of the weight blobs + a workspace; `forward(tokens)` launches the HIP pipeline on the current torch
    stream and returns a device tensor [B, D] fp32 (L2-normalised)."""

    def __init__(self, spec: XlmrSpec, sd: Dict[str, torch.Tensor], device: str = "cuda", max_batch: int = 8,
                 pack=None):
        """`spec`: anything with the XlmrSpec surface (`c_config`, context, vocab, pad_id, embed_dim, width);
        `pack(spec, sd)` -> (bf16 blob, fp32 blob), default pack_xlmr_weights"""
        self.spec = spec
        self.lib = _lib.lib()
        self.device = torch.device(device)
        self.cfg = spec.c_config()
        nb, nf = C.c_int64(), C.c_int64()
        _lib.check(self.lib.wise_xlmr_layout(C.byref(self.cfg), C.byref(nb), C.byref(nf)), "wise_xlmr_layout")
        wb, pf = (pack or pack_xlmr_weights)(spec, sd)
        if wb.numel() != nb.value or pf.numel() != nf.value:
            raise RuntimeError(f"weight blob size mismatch: packed {wb.numel()}/{pf.numel()}, "
                               f"library expects {nb.value}/{nf.value}")
        self.wb = wb.to(self.device)
        self.pf = pf.to(self.device)
        self._ws = None
        self._ws_batch = 0
        self.reserve(max_batch)
        # One query is ~175 launches of a few microseconds each (24 layers x 7 + head): launch-bound.  Small batches are
        # captured once into a hipGraph (the C ABI allocates and synchronises nothing) and replayed, as TextEngine does.
        self.graph_max_batch = 4
        self._graphs = {}

    def reserve(self, batch: int):
        if batch <= self._ws_batch:
            return
        n = self.lib.wise_xlmr_workspace_bytes(C.byref(self.cfg), batch)
        if n == 0:
            raise RuntimeError("wise_xlmr_workspace_bytes: bad config")
        self._ws = torch.empty(n, dtype=torch.uint8, device=self.device)
        self._ws_batch = batch
        self._graphs = {}  # captured graphs hold the old workspace address

    def _launch(self, t: torch.Tensor, out: torch.Tensor):
        _lib.check(self.lib.wise_xlmr_forward(C.byref(self.cfg), self.wb.data_ptr(), self.pf.data_ptr(), t.data_ptr(),
                                              t.shape[0], out.data_ptr(), self._ws.data_ptr(), self._ws.numel(),
                                              _lib.stream_ptr()), "wise_xlmr_forward")

    def _graph_for(self, B: int):
        hit = self._graphs.get(B)
        if hit is None:
            s = self.spec
            tok = torch.full((B, s.context), s.pad_id, dtype=torch.int32, device=self.device)
            tok[:, 0] = 0
            tok[:, 1] = 2
            out = torch.empty(B, s.embed_dim, dtype=torch.float32, device=self.device)
            self._launch(tok, out)  # warm-up outside the capture (first-call kernel attributes)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    self._launch(tok, out)
            except RuntimeError:
                self.graph_max_batch = 0   # capture not possible here: keep launching directly (same kernels)
                torch.cuda.synchronize()
                return None
            hit = (g, tok, out)
            self._graphs[B] = hit
        return hit

    def forward(self, tokens: torch.Tensor) -> torch.Tensor:
        s = self.spec
        if tokens.dim() != 2 or tokens.shape[1] != s.context or tokens.dtype not in (torch.int32, torch.int64):
            raise ValueError(f"expected integer tokens [B,{s.context}], got {tuple(tokens.shape)} {tokens.dtype}")
        if int(tokens.min()) < 0 or int(tokens.max()) >= s.vocab:
            raise ValueError("token id outside the vocabulary")
        live = tokens != s.pad_id
        n = live.sum(dim=1, keepdim=True)
        if bool((n == 0).any()) or not bool((live == (torch.arange(s.context, device=tokens.device)[None, :] < n)).all()):
            raise ValueError("token rows must be right-padded and non-empty (the HF tokenizer's padding='max_length')")
        t = tokens.to(device=self.device, dtype=torch.int32).contiguous()
        B = t.shape[0]
        self.reserve(B)
        if B <= self.graph_max_batch and not torch.cuda.is_current_stream_capturing():
            hit = self._graph_for(B)
            if hit is not None:
                g, tok, gout = hit
                tok.copy_(t)
                g.replay()
                return gout.clone()
        out = torch.empty(B, s.embed_dim, dtype=torch.float32, device=self.device)
        self._launch(t, out)
        return out

    def residual(self, batch: int) -> torch.Tensor:
        out = torch.empty(batch * self.spec.context, self.spec.width, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.wise_xlmr_tap_residual(C.byref(self.cfg), batch, self._ws.data_ptr(), out.data_ptr(),
                                                   _lib.stream_ptr()), "wise_xlmr_tap_residual")
        return out


def _clean(text: str) -> str:
    """open_clip tokenizer.py: whitespace_clean(basic_clean(text)) — ftfy is not a dependency this build carries, so
    basic_clean is html.unescape twice + strip (what ftfy.fix_text leaves of plain text)."""
    text = html.unescape(html.unescape(text)).strip()
    return re.sub(r"\s+", " ", text).strip()


class XlmrTokenizer:
    """open_clip's HFTokenizer('xlm-roberta-large') restated: clean -> sentencepiece pieces -> Hugging Face's fairseq
    alignment (<s>=0, <pad>=1, </s>=2, <unk>=3, every sentencepiece id shifted by one) -> `<s> ids </s>`, truncated to
    `context` with both specials kept, right-padded with <pad>.  Needs the model's `sentencepiece.bpe.model`."""

    BOS, PAD, EOS, UNK = 0, 1, 2, 3

    def __init__(self, model_file: Union[str, Path], context: int = 77):
        import sentencepiece as spm   # in the image; the only third-party piece of this tokenizer

        self.sp = spm.SentencePieceProcessor(model_file=str(model_file))
        self.context = int(context)
        self.vocab_size = self.sp.get_piece_size() + 2          # + the fairseq shift + <mask>

    @classmethod
    def default(cls, context: int = 77) -> "XlmrTokenizer":
        root = os.environ.get("WISE_AMD_WEIGHTS_DIR", "")
        path = Path(root) / "xlm-roberta-large" / "sentencepiece.bpe.model"
        if not root or not path.exists():
            raise FileNotFoundError("XLM-RoBERTa's sentencepiece.bpe.model not found: set WISE_AMD_WEIGHTS_DIR and place it at "
                                    "$WISE_AMD_WEIGHTS_DIR/xlm-roberta-large/sentencepiece.bpe.model")
        return cls(path, context)

    def encode(self, text: str) -> List[int]:
        ids = [i + 1 if i else self.UNK for i in self.sp.encode(_clean(text))]
        ids = ids[: self.context - 2]
        return [self.BOS] + ids + [self.EOS]

    def __call__(self, texts: Union[str, List[str]]) -> torch.Tensor:
        if isinstance(texts, str):
            texts = [texts]
        out = torch.full((len(texts), self.context), self.PAD, dtype=torch.int64)
        for r, t in enumerate(texts):
            ids = self.encode(t)
            out[r, : len(ids)] = torch.tensor(ids, dtype=torch.int64)
        return out
