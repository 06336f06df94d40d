"""Host side of the OpenCLIP text tower: model table, weight packing and the engine that
drives `wise_text_forward` (include/wise_hip.h).

Weights are addressed by open_clip 2.24.0 state-dict keys (`token_embedding.weight`, `positional_embedding`,
`transformer.resblocks.*`, `ln_final.*`, `text_projection`), the names `open_clip.create_model_and_transforms`
produces for the model the reference builds at src/feature/mlfoundation_openclip.py:38 and queries at :103-108.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Dict

import torch

from .. import _lib

SOT_TOKEN = 49406  # <start_of_text>
EOT_TOKEN = 49407  # <end_of_text>: the largest id, which is why open_clip pools at argmax(tokens)


@dataclass(frozen=True)
class TextSpec:
    name: str
    width: int
    heads: int
    layers: int
    embed_dim: int
    context: int = 77
    vocab: int = 49408
    act: str = "quick_gelu"   # 'quick_gelu' | 'gelu' | 'gelu_new' (GPT-2's tanh form)
    pool: str = "argmax"      # pooled row: 'argmax' of the ids (open_clip) | 'last_nonzero' id (msclap, pad id 0) | 'last' position
    head: str = "linear"      # 'linear' projection [W,D] | 'clap' = msclap Projection (W1, GELU, W2, LayerNorm) | 'linear_bias'
    causal: bool = True       # False: open_clip no_causal_mask (SigLIP)
    ln_eps: float = 1e-5      # 1e-6 for the SigLIP text tower (norm_kwargs)

    @property
    def mlp(self) -> int:
        return 4 * self.width

    def flops_per_query(self) -> int:
        T, W, F = self.context, self.width, self.mlp
        per_layer = T * W * 3 * W * 2 + 2 * self.heads * T * T * 64 * 2 + T * W * W * 2 + 2 * T * W * F * 2
        return self.layers * per_layer + W * self.embed_dim * 2

    def c_config(self) -> _lib.TextConfig:
        return _lib.TextConfig(self.context, self.vocab, self.width, self.layers, self.heads, self.mlp, self.embed_dim,
                               {"quick_gelu": 0, "gelu": 1, "gelu_new": 2}[self.act],
                               {"argmax": 0, "last_nonzero": 1, "last": 2}[self.pool],
                               {"linear": 0, "clap": 1, "linear_bias": 2}[self.head], 0 if self.causal else 1,
                               {1e-5: 0, 1e-6: 1}[self.ln_eps])


# text towers of the open_clip models in wise_amd/feature/vit.py:SPECS (open_clip model configs)
TEXT_SPECS: Dict[str, TextSpec] = {
    "ViT-B-32": TextSpec("ViT-B-32", 512, 8, 12, 512),
    "ViT-B-16": TextSpec("ViT-B-16", 512, 8, 12, 512),
    "ViT-L-14": TextSpec("ViT-L-14", 768, 12, 12, 768),
    "ViT-H-14": TextSpec("ViT-H-14", 1024, 16, 24, 1024),
}


def text_spec_for(model_name: str, pretrained: str = "openai") -> TextSpec:
    base = model_name.replace("-quickgelu", "")
    if base not in TEXT_SPECS:
        raise ValueError(f"Model ({model_name}, {pretrained}) not available")
    s = TEXT_SPECS[base]
    quick = model_name.endswith("-quickgelu") or pretrained == "openai"
    return TextSpec(**{**s.__dict__, "name": model_name, "act": "quick_gelu" if quick else "gelu"})


def text_state_dict_keys(spec: TextSpec):
    """(key, shape) in the order the seeded initialiser draws them."""
    W, F, D, T, V = spec.width, spec.mlp, spec.embed_dim, spec.context, spec.vocab
    keys = [("token_embedding.weight", (V, W)), ("positional_embedding", (T, W))]
    for i in range(spec.layers):
        p = f"transformer.resblocks.{i}."
        keys += [(p + "ln_1.weight", (W,)), (p + "ln_1.bias", (W,)), (p + "attn.in_proj_weight", (3 * W, W)),
                 (p + "attn.in_proj_bias", (3 * W,)), (p + "attn.out_proj.weight", (W, W)),
                 (p + "attn.out_proj.bias", (W,)), (p + "ln_2.weight", (W,)), (p + "ln_2.bias", (W,)),
                 (p + "mlp.c_fc.weight", (F, W)), (p + "mlp.c_fc.bias", (F,)), (p + "mlp.c_proj.weight", (W, F)),
                 (p + "mlp.c_proj.bias", (W,))]
    keys += [("ln_final.weight", (W,)), ("ln_final.bias", (W,)), ("text_projection", (W, D))]
    return keys


def random_text_state_dict(spec: TextSpec, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Seeded fp32 weights (no checkpoints exist offline); same conventions as vit.random_state_dict."""
    g = torch.Generator().manual_seed(1000 + seed)
    W, L = spec.width, max(spec.layers, 1)
    sd = {}
    for key, shape in text_state_dict_keys(spec):
        n = torch.randn(shape, generator=g, dtype=torch.float32)
        if key == "token_embedding.weight":
            t = n * 0.5
        elif key == "positional_embedding":
            t = n * 0.25
        elif key.endswith("ln_1.weight") or key.endswith("ln_2.weight") or key == "ln_final.weight":
            t = 1.0 + 0.1 * n
        elif ("ln_" in key) and key.endswith(".bias"):
            t = 0.1 * n
        elif key.endswith("in_proj_weight"):
            t = n * (W ** -0.5)
            t[: 2 * W] *= 2.0
        elif key.endswith("out_proj.weight"):
            t = n * (W ** -0.5) * ((2 * L) ** -0.5)
        elif key.endswith("c_fc.weight"):
            t = n * (W ** -0.5)
        elif key.endswith("c_proj.weight"):
            t = n * (spec.mlp ** -0.5) * ((2 * L) ** -0.5)
        elif key.endswith(".bias") or key.endswith("_bias"):
            t = 0.02 * n
        elif key == "text_projection":
            t = n * (W ** -0.5)
        else:
            raise KeyError(key)
        sd[key] = t.contiguous()
    return sd


def pack_text_weights(spec: TextSpec, sd: Dict[str, torch.Tensor]):
    """state dict -> (bf16 blob, fp32 blob) in the layout include/wise_hip.h documents (CPU tensors)."""
    f32 = lambda k: sd[k].detach().to(torch.float32).cpu()
    wb = []
    pf = [f32("token_embedding.weight").reshape(-1), f32("positional_embedding").reshape(-1)]
    for i in range(spec.layers):
        p = f"transformer.resblocks.{i}."
        wb += [f32(p + "attn.in_proj_weight").reshape(-1), f32(p + "attn.out_proj.weight").reshape(-1),
               f32(p + "mlp.c_fc.weight").reshape(-1), f32(p + "mlp.c_proj.weight").reshape(-1)]
        pf += [f32(p + "ln_1.weight"), f32(p + "ln_1.bias"), f32(p + "attn.in_proj_bias"),
               f32(p + "attn.out_proj.bias"), f32(p + "ln_2.weight"), f32(p + "ln_2.bias"), f32(p + "mlp.c_fc.bias"),
               f32(p + "mlp.c_proj.bias")]
    wb.append(f32("text_projection").t().contiguous().reshape(-1))  # text_projection^T [D, W]
    pf += [f32("ln_final.weight"), f32("ln_final.bias")]
    return torch.cat(wb).to(torch.bfloat16).contiguous(), torch.cat(pf).contiguous()


class TextEngine:
    """Device copies of the weight blobs + a workspace; `forward(tokens)` launches the HIP pipeline on the current
    torch stream and returns a device tensor [B, D] fp32 (L2-normalised)."""

    def __init__(self, spec: TextSpec, sd: Dict[str, torch.Tensor], device: str = "cuda", max_batch: int = 8,
                 pack=None):
        """`pack(spec, sd) -> (bf16 blob, fp32 blob)` defaults to the open_clip layout (pack_text_weights);
        the CLAP caption encoder passes its own (clap_text.pack_caption_weights)."""
        self.spec = spec
        self.lib = _lib.lib()
        self.device = torch.device(device)
        self.cfg = spec.c_config()
        nb, nf = C.c_int64(), C.c_int64()
        _lib.check(self.lib.wise_text_layout(C.byref(self.cfg), C.byref(nb), C.byref(nf)), "wise_text_layout")
        wb, pf = (pack or pack_text_weights)(spec, sd)
        if wb.numel() != nb.value or pf.numel() != nf.value:
            raise RuntimeError(f"weight blob size mismatch: packed {wb.numel()}/{pf.numel()}, "
                               f"library expects {nb.value}/{nf.value}")
        self.wb = wb.to(self.device)
        self.pf = pf.to(self.device)
        self._ws = None
        self._ws_batch = 0
        self.reserve(max_batch)
        # A single query is ~86 launches of a few microseconds each: launch-bound.  Small batches are therefore
        # captured once into a hipGraph (the C ABI allocates and synchronises nothing, so it is capturable) and
        # replayed; `graph_max_batch = 0` turns this off.
        self.graph_max_batch = 4
        self._graphs = {}

    def reserve(self, batch: int):
        if batch <= self._ws_batch:
            return
        n = self.lib.wise_text_workspace_bytes(C.byref(self.cfg), batch)
        if n == 0:
            raise RuntimeError("wise_text_workspace_bytes: bad config")
        self._ws = torch.empty(n, dtype=torch.uint8, device=self.device)
        self._ws_batch = batch
        self._graphs = {}  # captured graphs hold the old workspace address

    def _launch(self, t: torch.Tensor, out: torch.Tensor):
        rc = self.lib.wise_text_forward(C.byref(self.cfg), self.wb.data_ptr(), self.pf.data_ptr(), t.data_ptr(),
                                        t.shape[0], out.data_ptr(), self._ws.data_ptr(), self._ws.numel(),
                                        _lib.stream_ptr())
        _lib.check(rc, "wise_text_forward")

    def _graph_for(self, B: int):
        hit = self._graphs.get(B)
        if hit is None:
            tok = torch.zeros(B, self.spec.context, dtype=torch.int32, device=self.device)
            out = torch.empty(B, self.spec.embed_dim, dtype=torch.float32, device=self.device)
            self._launch(tok, out)  # warm-up outside the capture (first-call kernel attributes)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            try:
                # thread-local capture mode: other threads of the process (e.g. the RCCL watchdog) may call the
                # runtime while this thread captures
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
        if tokens.dim() != 2 or tokens.shape[1] != self.spec.context or tokens.dtype not in (torch.int32, torch.int64):
            raise ValueError(f"expected integer tokens [B,{self.spec.context}], got {tuple(tokens.shape)} {tokens.dtype}")
        if int(tokens.min()) < 0 or int(tokens.max()) >= self.spec.vocab:
            raise ValueError("token id outside the vocabulary")
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
        out = torch.empty(B, self.spec.embed_dim, dtype=torch.float32, device=self.device)
        self._launch(t, out)
        return out

    def residual(self, batch: int) -> torch.Tensor:
        out = torch.empty(batch * self.spec.context, self.spec.width, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.wise_text_tap_residual(C.byref(self.cfg), batch, self._ws.data_ptr(), out.data_ptr(),
                                                   _lib.stream_ptr()), "wise_text_tap_residual")
        return out
