"""Host side of the OpenCLIP image tower: model table, weight packing, and the engine that drives
`wise_vit_forward` (include/wise_hip.h).

Weights are addressed by open_clip 2.24.0 state-dict keys (the names the reference's
`open_clip.create_model_and_transforms` produces, src/feature/mlfoundation_openclip.py:38), so a
real checkpoint is a pure data problem: `pack_weights(spec, state_dict)`.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Dict, Optional

import torch

from .. import _lib


@dataclass(frozen=True)
class VitSpec:
    name: str
    image_size: int
    patch: int
    width: int
    layers: int
    heads: int
    mlp: int
    embed_dim: int
    act: str = "quick_gelu"  # 'quick_gelu' (openai tags) | 'gelu' (laion tags) | 'gelu_tanh'
    arch: int = 0            # 0 = open_clip VisionTransformer; 1 = timm ViT of the SigLIP towers (wise_amd/feature/siglip.py)
    ln_fold: int = 0         # 1: the blocks' LayerNorms folded into the GEMMs around them (wise_vit_config.ln_fold; pack_weights);
                             # 2: that, with attention + out-projection + residual add as one kernel (<= 64 tokens, 12 heads of 64)

    @property
    def grid(self) -> int:
        return self.image_size // self.patch

    @property
    def tokens(self) -> int:
        return self.grid * self.grid + (1 if self.arch == 0 else 0)

    @property
    def kdim(self) -> int:
        return 3 * self.patch * self.patch

    @property
    def kpad(self) -> int:
        return (self.kdim + 63) // 64 * 64

    def flops_per_frame(self) -> int:
        """2*MAC of the contractions only."""
        g2, T, W, F, D = self.grid ** 2, self.tokens, self.width, self.mlp, self.embed_dim
        per_layer = T * W * 3 * W * 2 + 2 * self.heads * T * T * (W // self.heads) * 2 + T * W * W * 2 + 2 * T * W * F * 2
        if self.arch == 1:   # attention-pool head: kv over every token, one query, projection, MLP on the pooled row
            return g2 * self.kdim * W * 2 + self.layers * per_layer + T * W * 2 * W * 2 + 2 * T * W * 2 + W * W * 2 + 2 * W * F * 2
        return g2 * self.kdim * W * 2 + self.layers * per_layer + W * D * 2

    def c_config(self) -> _lib.VitConfig:
        return _lib.VitConfig(self.image_size, self.patch, self.width, self.layers, self.heads, self.mlp,
                              self.embed_dim, {"quick_gelu": 0, "gelu": 1, "gelu_tanh": 2}[self.act], self.arch,
                              int(self.ln_fold))


# open_clip model names WISE passes as id token [2]
SPECS: Dict[str, VitSpec] = {
    "ViT-B-32": VitSpec("ViT-B-32", 224, 32, 768, 12, 12, 3072, 512),
    "ViT-B-16": VitSpec("ViT-B-16", 224, 16, 768, 12, 12, 3072, 512),
    "ViT-L-14": VitSpec("ViT-L-14", 224, 14, 1024, 24, 16, 4096, 768),
    # head width 80 (open_clip model_configs/ViT-H-14.json); xlm-roberta-large-ViT-H-14 has the same image tower
    "ViT-H-14": VitSpec("ViT-H-14", 224, 14, 1280, 32, 16, 5120, 1024),
    "xlm-roberta-large-ViT-H-14": VitSpec("xlm-roberta-large-ViT-H-14", 224, 14, 1280, 32, 16, 5120, 1024),
    "xlm-roberta-base-ViT-B-32": VitSpec("xlm-roberta-base-ViT-B-32", 224, 32, 768, 12, 12, 3072, 512),
}


def spec_for(model_name: str, pretrained: str = "openai") -> VitSpec:
    base = model_name.replace("-quickgelu", "")
    if base not in SPECS:
        raise ValueError(f"Model ({model_name}, {pretrained}) not available")
    s = SPECS[base]
    quick = model_name.endswith("-quickgelu") or pretrained == "openai"
    return VitSpec(**{**s.__dict__, "name": model_name, "act": "quick_gelu" if quick else "gelu"})


def state_dict_keys(spec: VitSpec):
    """(key, shape) in the order the seeded initialiser draws them."""
    W, F, D, T, P = spec.width, spec.mlp, spec.embed_dim, spec.tokens, spec.patch
    keys = [("visual.conv1.weight", (W, 3, P, P)), ("visual.class_embedding", (W,)),
            ("visual.positional_embedding", (T, W)), ("visual.ln_pre.weight", (W,)), ("visual.ln_pre.bias", (W,))]
    for i in range(spec.layers):
        p = f"visual.transformer.resblocks.{i}."
        keys += [(p + "ln_1.weight", (W,)), (p + "ln_1.bias", (W,)), (p + "attn.in_proj_weight", (3 * W, W)),
                 (p + "attn.in_proj_bias", (3 * W,)), (p + "attn.out_proj.weight", (W, W)),
                 (p + "attn.out_proj.bias", (W,)), (p + "ln_2.weight", (W,)), (p + "ln_2.bias", (W,)),
                 (p + "mlp.c_fc.weight", (F, W)), (p + "mlp.c_fc.bias", (F,)), (p + "mlp.c_proj.weight", (W, F)),
                 (p + "mlp.c_proj.bias", (W,))]
    keys += [("visual.ln_post.weight", (W,)), ("visual.ln_post.bias", (W,)), ("visual.proj", (W, D))]
    return keys


def random_state_dict(spec: VitSpec, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Seeded fp32 weights (no checkpoints exist offline).  One CPU generator, tensors drawn in
    `state_dict_keys` order; LayerNorm affine terms are perturbed so a dropped scale/shift shows;
    q/k projections are scaled up so attention is far from uniform."""
    g = torch.Generator().manual_seed(seed)
    W, L = spec.width, max(spec.layers, 1)
    sd = {}
    for key, shape in state_dict_keys(spec):
        n = torch.randn(shape, generator=g, dtype=torch.float32)
        if key.endswith("conv1.weight"):
            t = n * (spec.kdim ** -0.5)
        elif key.endswith("ln_pre.weight") or key.endswith("ln_1.weight") or key.endswith("ln_2.weight") \
                or key.endswith("ln_post.weight"):
            t = 1.0 + 0.1 * n
        elif ".ln_" in key and key.endswith(".bias"):
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
        elif key.endswith("class_embedding") or key.endswith("positional_embedding"):
            t = n * 0.5
        elif key.endswith("visual.proj"):
            t = n * (W ** -0.5)
        else:
            raise KeyError(key)
        sd[key] = t.contiguous()
    return sd


def checkpoint_like_state_dict(spec: VitSpec, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Seeded weights with the statistics that make real CLIP checkpoints hard for a bf16 path (no checkpoint exists
    offline; the benign Gaussians of `random_state_dict` do not show these):
      * massive activations — four residual channels carry values 40-100x the typical magnitude from the embeddings on
        (class / positional embedding entries of +-60..100) and a fifth is switched on by the c_proj bias of block 1;
      * heavy-tailed LayerNorm gains — log-normal (sigma 0.6, clamped to [0.05, 12]); on the massive channels the gains
        alternate between tiny (0.05: the channel is squashed, as trained models do) and large (3.0: the massive value
        reaches the next GEMM's bf16 operand at full size);
      * large embeddings — class and positional embeddings at 3x the benign scale;
      * peaky attention — the q and k projections of the first third of the heads are scaled so that their logits are
        ~9x the benign ones (near one-hot softmax rows).
    Same key order and generator discipline as `random_state_dict` (the GPU box regenerates the weights from the seed).
    Parity on these weights is what tests/test_gpu_vit.py::test_vit_checkpoint_like_golden holds the HIP path to."""
    sd = random_state_dict(spec, seed)
    g = torch.Generator().manual_seed(seed + 7919)
    W, H = spec.width, spec.heads
    dh = W // H
    chans = torch.randperm(W, generator=g)[:5].tolist()
    big = [80.0, -60.0, 100.0, -70.0]
    sd["visual.class_embedding"] = sd["visual.class_embedding"] * 3.0
    sd["visual.positional_embedding"] = sd["visual.positional_embedding"] * 3.0
    for c, v in zip(chans[:4], big):
        sd["visual.class_embedding"][c] = v
        sd["visual.positional_embedding"][:, c] = 0.5 * v
    ln_keys = [k for k in sd if (".ln_" in k or "ln_pre" in k or "ln_post" in k) and k.endswith(".weight")]
    for n, key in enumerate(ln_keys):
        w = torch.exp(0.6 * torch.randn(sd[key].shape, generator=g)).clamp_(0.05, 12.0)
        for c in chans:
            w[c] = 0.05 if n % 2 == 0 else 3.0
        sd[key] = w.contiguous()
    if spec.layers >= 2:
        sd["visual.transformer.resblocks.1.mlp.c_proj.bias"][chans[4]] = 90.0
    hot = max(1, H // 3)
    for l in range(spec.layers):
        key = f"visual.transformer.resblocks.{l}.attn.in_proj_weight"
        w = sd[key]
        w[: hot * dh] *= 3.0                 # q rows of the hot heads
        w[W: W + hot * dh] *= 3.0            # k rows of the hot heads
    return sd


def fold_layernorm(weight: torch.Tensor, bias: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor):
    """Linear(LayerNorm(x)) with the LayerNorm's affine part and its centring moved into the linear layer:
    LN(x) W^T + b = rstd * (x W''^T) + b'' with W'' = gamma * W minus its mean over the input dimension (so that
    sum_k (x_k - mean(x)) w_k = sum_k x_k (w_k - mean_k(w)): the kernel never needs mean(x)) and b'' = b + W beta.
    What is left for the GEMM's epilogue is one scale per row, rstd = 1 / sqrt(var(x) + eps) (gemm_w4.h FoldArgs)."""
    wg = weight.double() * gamma.double()[None, :]
    wg = wg - wg.mean(dim=1, keepdim=True)
    return wg.to(torch.float32), (bias.double() + weight.double() @ beta.double()).to(torch.float32)


def pack_weights(spec: VitSpec, sd: Dict[str, torch.Tensor]):
    """state dict -> (bf16 blob, fp32 blob) in the layout include/wise_hip.h documents (CPU tensors).  With spec.ln_fold the
    in_proj / c_fc slots hold the folded weights and biases (fold_layernorm); the ln_1 / ln_2 slots stay as they are (unread);
    with spec.ln_fold == 2 the out_proj slot holds W_o in tile order (tile_out_proj)."""
    W, F = spec.width, spec.mlp
    f32 = lambda k: sd[k].detach().to(torch.float32).cpu()
    conv = torch.zeros(W, spec.kpad, dtype=torch.float32)
    conv[:, : spec.kdim] = f32("visual.conv1.weight").reshape(W, spec.kdim)
    wb = [conv.reshape(-1)]
    pf = [f32("visual.class_embedding").reshape(-1), f32("visual.positional_embedding").reshape(-1),
          f32("visual.ln_pre.weight"), f32("visual.ln_pre.bias")]
    for i in range(spec.layers):
        p = f"visual.transformer.resblocks.{i}."
        w_in, b_in = f32(p + "attn.in_proj_weight"), f32(p + "attn.in_proj_bias")
        w_fc, b_fc = f32(p + "mlp.c_fc.weight"), f32(p + "mlp.c_fc.bias")
        if spec.ln_fold:
            w_in, b_in = fold_layernorm(w_in, b_in, f32(p + "ln_1.weight"), f32(p + "ln_1.bias"))
            w_fc, b_fc = fold_layernorm(w_fc, b_fc, f32(p + "ln_2.weight"), f32(p + "ln_2.bias"))
        w_out = f32(p + "attn.out_proj.weight")
        if spec.ln_fold == 2:       # the one-kernel attention half streams W_o in tile order
            w_out = tile_out_proj(w_out)
        wb += [w_in.reshape(-1), w_out.reshape(-1), w_fc.reshape(-1),
               f32(p + "mlp.c_proj.weight").reshape(-1)]
        pf += [f32(p + "ln_1.weight"), f32(p + "ln_1.bias"), b_in, f32(p + "attn.out_proj.bias"), f32(p + "ln_2.weight"),
               f32(p + "ln_2.bias"), b_fc, f32(p + "mlp.c_proj.bias")]
    wb.append(f32("visual.proj").t().contiguous().reshape(-1))  # proj^T [D, W]
    pf += [f32("visual.ln_post.weight"), f32("visual.ln_post.bias")]
    wb_t = torch.cat(wb).to(torch.bfloat16).contiguous()
    pf_t = torch.cat(pf).contiguous()
    return wb_t, pf_t


class PendingEmbeddings:
    """Handle of a batch enqueued by VitEngine.forward_pipelined."""

    def __init__(self, out: torch.Tensor, done: "torch.cuda.Event"):
        self._out, self._done = out, done

    @property
    def device(self) -> torch.device:
        return self._out.device

    def result(self) -> torch.Tensor:
        """The embeddings [B, D], ordered after the batch on the caller's current stream (no host sync)."""
        torch.cuda.current_stream(self._out.device).wait_event(self._done)
        return self._out


def tile_out_proj(w: torch.Tensor) -> torch.Tensor:
    """W_o [W, W] (out, in) -> the same elements in the order wise_attention_oproj_fold streams them (include/wise_hip.h):
    [wave = out // 64][K-step = in // 64][k-half][column tile j][lane = 16 * g + l][8], the element being
    W_o[64 * wave + 16 * j + l, 64 * step + 32 * half + 8 * g + e] — a (wave, step, half, j) fragment is 1 KiB contiguous."""
    W = w.shape[0]
    assert w.shape == (W, W) and W % 64 == 0
    t = w.reshape(W // 64, 4, 16, W // 64, 2, 4, 8)          # wave, j, l, step, half, g, e
    return t.permute(0, 3, 4, 1, 5, 2, 6).contiguous().reshape(W, W)


DEFAULT_FOLD_B32 = 1   # what VitEngine picks for the ViT-B/32 shape (see its ln_fold argument)


class VitEngine:
    """Owns the device copies of the two weight blobs and a workspace; `forward` launches the HIP
    pipeline on the current torch stream and returns a device tensor [B, D] fp32 (L2-normalised)."""

    def __init__(self, spec: VitSpec, sd: Dict[str, torch.Tensor], device: str = "cuda", max_batch: int = 256,
                 ln_fold: Optional[int] = None):
        """ln_fold: fold the blocks' LayerNorms into the GEMMs around them and keep the residual stream as bf16 hi + lo (two
        launches and a whole pass over the rows fewer per LayerNorm; results within the same tolerance of the fp32 path, not
        bit-equal to the unfolded form).  None = WISE_VIT_LN_FOLD (0 / 1); default: on where it measured faster — ViT-B/32
        (width 768, 50 tokens: bs=256 two batches in flight 2.85 -> 2.57 ms per step, bs=8 .. 64 one stream 9 - 15 % faster) —
        and off where it measured slower: B/16 (-2 %), L/14 (-2.5 %), H/14 (-3 %), whose row counts are no multiple of the 160-row
        tiles, so that the folded GEMMs fall to 128-row tiles (tools/vit_fold_ab.py, profiles/r04_fold_ab.txt).  Never for
        the timm towers (arch 1)."""
        fused_ok = spec.heads == 12 and spec.width == 768 and spec.tokens <= 64
        if ln_fold is None:
            env = os.environ.get("WISE_VIT_LN_FOLD", "")
            ln_fold = int(env) if env in ("0", "1", "2") else (DEFAULT_FOLD_B32 if spec.arch == 0 and fused_ok else 0)
        ln_fold = int(ln_fold) if (spec.arch == 0 and spec.layers >= 1 and spec.width >= 256) else 0
        if ln_fold not in (0, 1, 2) or (ln_fold == 2 and not fused_ok):
            raise ValueError(f"ln_fold={ln_fold}: 0, 1, or 2 (2: up to 64 tokens and 12 heads of 64 only)")
        if ln_fold != spec.ln_fold:
            spec = VitSpec(**{**spec.__dict__, "ln_fold": ln_fold})
        self.spec = spec
        self.lib = _lib.lib()
        self.device = torch.device(device)
        self.cfg = spec.c_config()
        nb, nf = C.c_int64(), C.c_int64()
        _lib.check(self.lib.wise_vit_layout(C.byref(self.cfg), C.byref(nb), C.byref(nf)), "wise_vit_layout")
        if spec.arch == 1:
            from .siglip import pack_siglip_vision
            wb, pf = pack_siglip_vision(spec, sd)
        else:
            wb, pf = pack_weights(spec, sd)
        if wb.numel() != nb.value or pf.numel() != nf.value:
            raise RuntimeError(f"weight blob size mismatch: packed {wb.numel()}/{pf.numel()}, "
                               f"library expects {nb.value}/{nf.value}")
        self.wb = wb.to(self.device)
        self.pf = pf.to(self.device)
        self._ws = None
        self._ws_batch = 0
        self.reserve(max_batch)

    def reserve(self, batch: int):
        if batch <= self._ws_batch:
            return
        n = self.lib.wise_vit_workspace_bytes(C.byref(self.cfg), batch)
        if n == 0:
            raise RuntimeError("wise_vit_workspace_bytes: bad config")
        self._ws = torch.empty(n, dtype=torch.uint8, device=self.device)
        self._ws_batch = batch

    def _check_images(self, images: torch.Tensor):
        S = self.spec.image_size
        if images.dim() != 4 or tuple(images.shape[1:]) != (3, S, S):
            raise ValueError(f"expected [B,3,{S},{S}], got {tuple(images.shape)}")
        if images.dtype == torch.uint8:
            kind = _lib.WISE_VIT_IN_U8
        elif images.dtype == torch.float32:
            kind = _lib.WISE_VIT_IN_F32
        else:
            raise ValueError(f"images must be float32 or uint8, got {images.dtype}")
        return images.to(self.device).contiguous(), kind

    def forward(self, images: torch.Tensor, single_stream: bool = False) -> torch.Tensor:
        """One batch -> [B, D] fp32 unit rows on the current torch stream.  By default the library overlaps two half
        batches on two streams (wise_vit_forward); single_stream=True keeps every launch on the caller's stream
        (wise_vit_forward_single: what a profiler or an event-bracketed measurement wants)."""
        x, kind = self._check_images(images)
        B = x.shape[0]
        self.reserve(B)
        out = torch.empty(B, self.spec.embed_dim, dtype=torch.float32, device=self.device)
        fn = self.lib.wise_vit_forward_single if single_stream else self.lib.wise_vit_forward
        rc = fn(C.byref(self.cfg), self.wb.data_ptr(), self.pf.data_ptr(), x.data_ptr(), kind, B,
                out.data_ptr(), self._ws.data_ptr(), self._ws.numel(), _lib.stream_ptr())
        _lib.check(rc, "wise_vit_forward")
        return out

    # -- two whole batches in flight ---------------------------------------------------------------
    def forward_pipelined(self, images: torch.Tensor) -> "PendingEmbeddings":
        """Enqueue one batch and return at once; `.result()` of the returned handle gives the embeddings.

        Successive calls alternate between two slots, each with its own stream and workspace, so the GPU always
        holds two whole batches: a batch runs with full-batch GEMM shapes (which tile the chip better than the
        half batches of `forward`) and the other batch's kernels fill its LayerNorm / attention phases and tails.
        Use it where batches arrive in a stream (extract-features style loops) and a result is consumed one batch
        later; `forward` stays the call for a single batch."""
        x, kind = self._check_images(images)
        B = x.shape[0]
        if not hasattr(self, "_slots"):
            self._slots, self._next_slot = [], 0
        need = self.lib.wise_vit_workspace_bytes(C.byref(self.cfg), B)
        nslots = getattr(self, "pipeline_depth", 2)
        if len(self._slots) < nslots:
            from .._streams import concurrent_streams   # streams SEEN to run side by side (two on one hardware queue: no overlap)
            if self._slots:
                torch.cuda.synchronize(self.device)      # (depth raised on a live engine: nothing may still run in the old slots)
            self._slots = [{"stream": st, "ws": None} for st in concurrent_streams(nslots, self.device)]
            self._next_slot = 0
        slot = self._slots[self._next_slot]
        self._next_slot = (self._next_slot + 1) % nslots
        if slot["ws"] is None or slot["ws"].numel() < need:
            # the old workspace may still be in use by this slot's previous forward, and it was allocated on another
            # stream than the one that uses it: wait for that forward before the allocator may hand the block out again
            slot["stream"].synchronize()
            slot["ws"] = torch.empty(need, dtype=torch.uint8, device=self.device)
        cur = torch.cuda.current_stream(self.device)
        slot["stream"].wait_stream(cur)              # the batch was produced on the caller's stream
        out = torch.empty(B, self.spec.embed_dim, dtype=torch.float32, device=self.device)
        x.record_stream(slot["stream"])
        out.record_stream(slot["stream"])
        self.lib.wise_overlap_hint(1)   # this batch runs beside the other slot's: GEMM tiles chosen for co-residency
        rc = self.lib.wise_vit_forward_single(C.byref(self.cfg), self.wb.data_ptr(), self.pf.data_ptr(), x.data_ptr(),
                                              kind, B, out.data_ptr(), slot["ws"].data_ptr(), slot["ws"].numel(),
                                              slot["stream"].cuda_stream)
        self.lib.wise_overlap_hint(0)
        _lib.check(rc, "wise_vit_forward_single")
        done = torch.cuda.Event()
        done.record(slot["stream"])
        return PendingEmbeddings(out, done)

    def residual(self, batch: int) -> torch.Tensor:
        """Residual stream [B*T, W] fp32 left by the last forward (parity tap)."""
        out = torch.empty(batch * self.spec.tokens, self.spec.width, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.wise_vit_tap_residual(C.byref(self.cfg), batch, self._ws.data_ptr(), out.data_ptr(),
                                                  _lib.stream_ptr()), "wise_vit_tap_residual")
        return out
