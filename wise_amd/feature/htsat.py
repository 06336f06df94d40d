"""Host side of the MS-CLAP (version 2023) HTSAT audio encoder: state-dict layout, seeded initialiser,
weight packing, and the engine that drives `wise_htsat_forward` (include/wise_hip.h).

Weights are addressed by msclap 1.3.3 state-dict keys under `clap.audio_encoder.` (what
`msclap.CLAP(version='2023')` loads, reference call site src/feature/microsoft_clap.py:31), so a real
checkpoint is a pure data problem.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Tuple

import torch

from .. import _lib

N_FFT, HOP, N_MELS = 1024, 320, 64
SPEC_SIZE, FREQ_RATIO, WINDOW = 256, 4, 8
DEPTHS = (2, 2, 6, 2)
HEADS = (4, 8, 16, 32)
EMBED = 96
LATENT = 768
OUT_DIM = 1024
MAX_FRAMES = SPEC_SIZE * FREQ_RATIO  # 1024


def state_dict_keys() -> List[Tuple[str, Tuple[int, ...]]]:
    """(key, shape) in the order the seeded initialiser draws them (front-end buffers are not weights:
    the STFT kernels / mel filterbank are fixed functions of the config and are rebuilt, not loaded)."""
    pre = "base.htsat."
    keys = [(pre + "bn0.weight", (N_MELS,)), (pre + "bn0.bias", (N_MELS,)), (pre + "bn0.running_mean", (N_MELS,)),
            (pre + "bn0.running_var", (N_MELS,)), (pre + "patch_embed.proj.weight", (EMBED, 1, 4, 4)),
            (pre + "patch_embed.proj.bias", (EMBED,)), (pre + "patch_embed.norm.weight", (EMBED,)),
            (pre + "patch_embed.norm.bias", (EMBED,))]
    for i, depth in enumerate(DEPTHS):
        Cd = EMBED << i
        for j in range(depth):
            p = f"{pre}layers.{i}.blocks.{j}."
            keys += [(p + "norm1.weight", (Cd,)), (p + "norm1.bias", (Cd,)),
                     (p + "attn.relative_position_bias_table", ((2 * WINDOW - 1) ** 2, HEADS[i])),
                     (p + "attn.qkv.weight", (3 * Cd, Cd)), (p + "attn.qkv.bias", (3 * Cd,)),
                     (p + "attn.proj.weight", (Cd, Cd)), (p + "attn.proj.bias", (Cd,)),
                     (p + "norm2.weight", (Cd,)), (p + "norm2.bias", (Cd,)),
                     (p + "mlp.fc1.weight", (4 * Cd, Cd)), (p + "mlp.fc1.bias", (4 * Cd,)),
                     (p + "mlp.fc2.weight", (Cd, 4 * Cd)), (p + "mlp.fc2.bias", (Cd,))]
        if i < 3:
            p = f"{pre}layers.{i}.downsample."
            keys += [(p + "norm.weight", (4 * Cd,)), (p + "norm.bias", (4 * Cd,)), (p + "reduction.weight", (2 * Cd, 4 * Cd))]
    keys += [(pre + "norm.weight", (LATENT,)), (pre + "norm.bias", (LATENT,)),
             ("projection.linear1.weight", (OUT_DIM, LATENT)), ("projection.linear2.weight", (OUT_DIM, OUT_DIM)),
             ("projection.layer_norm.weight", (OUT_DIM,)), ("projection.layer_norm.bias", (OUT_DIM,))]
    return keys


def random_htsat_state_dict(seed: int = 0) -> Dict[str, torch.Tensor]:
    """Seeded fp32 weights (no checkpoint exists offline); one CPU generator, `state_dict_keys` order."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for key, shape in state_dict_keys():
        n = torch.randn(shape, generator=g, dtype=torch.float32)
        fan_in = shape[-1] if len(shape) == 2 else 16
        if key.endswith("bn0.running_mean"):
            t = -30.0 + 5.0 * n            # log-mel dB of 0.1-amplitude noise sits around -30 dB
        elif key.endswith("bn0.running_var"):
            t = 100.0 * (1.0 + 0.2 * n).abs() + 1.0
        elif key.endswith("norm.weight") or key.endswith("norm1.weight") or key.endswith("norm2.weight") \
                or key.endswith("bn0.weight") or key.endswith("layer_norm.weight"):
            t = 1.0 + 0.1 * n
        elif key.endswith("relative_position_bias_table"):
            t = 0.5 * n
        elif key.endswith("patch_embed.proj.weight"):
            t = n * 0.25
        elif key.endswith("qkv.weight"):
            t = n * (fan_in ** -0.5)
            t[: 2 * shape[1]] *= 1.5      # sharper attention than a default init gives
        elif key.endswith(".weight") and len(shape) == 2:
            t = n * (fan_in ** -0.5) * (0.5 if ("fc2" in key or "attn.proj" in key) else 1.0)
        elif key.endswith(".bias"):
            t = (0.1 if "norm" in key or "bn0" in key else 0.02) * n
        else:
            raise KeyError(key)
        sd[key] = t.contiguous()
    return sd


def checkpoint_like_htsat_state_dict(seed: int = 0) -> Dict[str, torch.Tensor]:
    """Seeded weights with the statistics of trained Swin / HTSAT checkpoints that the benign Gaussians above lack (no
    checkpoint exists offline): log-normal LayerNorm gains (sigma 0.6, clamped to [0.05, 12]); three massive channels per
    stage switched on by fc2 biases of +-40..80 in the stage's first block (and squashed or amplified by the norms that
    follow, alternately); a relative-position-bias table with a few entries at +-6 (near one-hot window attention); q / k
    projections of the first third of the heads scaled 3x.  Same key order as `random_htsat_state_dict`.
    tests/test_gpu_htsat.py::test_checkpoint_like_golden holds the HIP path to cosine >= 1 - 1e-3 on them."""
    sd = random_htsat_state_dict(seed)
    g = torch.Generator().manual_seed(seed + 104729)
    pre = "base.htsat."
    for i, depth in enumerate(DEPTHS):
        Cd = EMBED << i
        dh = Cd // HEADS[i]
        chans = torch.randperm(Cd, generator=g)[:3].tolist()
        vals = [60.0, -40.0, 80.0]
        for c, v in zip(chans, vals):
            sd[f"{pre}layers.{i}.blocks.0.mlp.fc2.bias"][c] = v
        n_ln = 0
        for j in range(depth):
            p = f"{pre}layers.{i}.blocks.{j}."
            for nm in ("norm1.weight", "norm2.weight"):
                w = torch.exp(0.6 * torch.randn(Cd, generator=g)).clamp_(0.05, 12.0)
                for c in chans:
                    w[c] = 0.05 if n_ln % 2 == 0 else 3.0
                n_ln += 1
                sd[p + nm] = w.contiguous()
            tab = sd[p + "attn.relative_position_bias_table"]
            hot_rows = torch.randperm(tab.shape[0], generator=g)[:6]
            tab[hot_rows, :] = 6.0 * torch.sign(torch.randn(6, tab.shape[1], generator=g))
            hot = max(1, HEADS[i] // 3)
            sd[p + "attn.qkv.weight"][: hot * dh] *= 3.0
            sd[p + "attn.qkv.weight"][Cd: Cd + hot * dh] *= 3.0
    sd[pre + "norm.weight"] = torch.exp(0.6 * torch.randn(LATENT, generator=g)).clamp_(0.05, 12.0)
    return sd


DEFAULT_ATTN_STREAM = True   # what HtsatEngine picks for attn_stream=None
DEFAULT_MLP_STREAM = True    # what HtsatEngine picks for mlp_stream=None (see its docstring)


class HtsatEngine:
    """Device copies of the packed weights + workspace; forward(wave [B,N] fp32) -> [B,1024] fp32 device
    tensor, L2-normalised (microsoft_clap.py:49-50)."""

    def __init__(self, sd: Dict[str, torch.Tensor], device: str = "cuda", max_batch: int = 128,
                 max_samples: int = 480000, ln_fold=None, mlp_stream=None, attn_stream=None):
        """ln_fold: stages 2 - 4 with their LayerNorms folded into the GEMMs around them and the residual stream as bf16
        hi + lo (wise_htsat_forward2 flags bit 0; same tolerance of the fp32 path, not bit-equal to the unfolded form).
        None = WISE_HTSAT_LN_FOLD (0 / 1), default OFF: built, parity-green and measured no faster here (bs=128 x 10 s: 3.53 ->
        3.58 ms one batch at a time, 3.28 -> 3.28 ms with two in flight; profiles/r04_htsat_fold_ab.txt — its first
        measurement, 3.33 -> 3.67 ms in flight, was an artefact: that engine's two streams shared a hardware queue, see
        wise_amd/_streams.py).  A stage-2 / -3 block
        does lose its two LayerNorm launches (-34 / -10 us), but these stages are bound by the bytes of their own operands
        (K = 192 / 384: three to six K-steps per tile), and the fold's epilogue — hi + lo join and split, the statistics'
        tree — is ~45 vector instructions per 16 bytes in a kernel with one wave per SIMD: it stops hiding under the store
        burst (projection 23.8 -> 35.0 us, patch-merging GEMMs +14 .. +30 us), and beside a second batch it holds the CU.

        mlp_stream: the MLP of every block of stages 2 and 3 as one kernel whose 4C-wide hidden activations never reach HBM
        (wise_mlp_stream; wise_htsat_forward2 flags bit 1; the fc1 / fc2 slots then hold that kernel's weight stream).
        None = WISE_HTSAT_MLP_STREAM (0 / 1), default ON: bs=128 x 10 s 3.55 -> 3.40 ms one batch at a time, 3.33 -> 3.21 ms
        with two in flight (38.4 k -> 40.0 k clips/s; profiles/r04_mlp_stream_study.txt).  Not together with ln_fold.

        attn_stream: norm1 + QKV projection + window attention of every block of stages 2 and 3 as one kernel (wise_swin_qkv_attn,
        flags bit 2; the qkv slots then hold that kernel's stream).  None = WISE_HTSAT_ATTN_STREAM (0 / 1), default ON: bs=128 x
        10 s 3.46 -> 3.31 ms one batch at a time, 3.13 -> 3.10 ms with two in flight (profiles/r04_swin_stream_study.txt).  Not
        together with ln_fold."""
        self.lib = _lib.lib()
        self.device = torch.device(device)
        if ln_fold is None:
            ln_fold = os.environ.get("WISE_HTSAT_LN_FOLD", "0") == "1"
        self.ln_fold = bool(ln_fold)
        if mlp_stream is None:
            env = os.environ.get("WISE_HTSAT_MLP_STREAM", "")
            mlp_stream = (env == "1") if env in ("0", "1") else DEFAULT_MLP_STREAM
        self.mlp_stream = bool(mlp_stream) and not self.ln_fold
        if attn_stream is None:
            env = os.environ.get("WISE_HTSAT_ATTN_STREAM", "")
            attn_stream = (env == "1") if env in ("0", "1") else DEFAULT_ATTN_STREAM
        self.attn_stream = bool(attn_stream) and not self.ln_fold
        self._flags = (1 if self.ln_fold else 0) | (2 if self.mlp_stream else 0) | (4 if self.attn_stream else 0)
        nb, nf = C.c_int64(), C.c_int64()
        _lib.check(self.lib.wise_htsat_layout(C.byref(nb), C.byref(nf)), "wise_htsat_layout")
        wb, pf = pack_htsat_weights(sd, fold=self.ln_fold, mlp_stream=self.mlp_stream, attn_stream=self.attn_stream)
        if wb.numel() != nb.value or pf.numel() != nf.value:
            raise RuntimeError(f"HTSAT blob size mismatch: packed {wb.numel()}/{pf.numel()}, "
                               f"library expects {nb.value}/{nf.value}")
        self.wb, self.pf = wb.to(self.device), pf.to(self.device)
        self._ws = None
        self._ws_key = (0, 0)
        self._last = (0, 0)
        self.reserve(max_batch, max_samples)

    def reserve(self, batch: int, samples: int):
        if batch <= self._ws_key[0] and samples <= self._ws_key[1]:
            return
        batch, samples = max(batch, self._ws_key[0]), max(samples, self._ws_key[1])
        n = self.lib.wise_htsat_workspace_bytes(batch, samples)
        if n == 0:
            raise RuntimeError("wise_htsat_workspace_bytes: bad shape")
        self._ws = torch.empty(n, dtype=torch.uint8, device=self.device)
        self._ws_key = (batch, samples)

    def forward(self, wave: torch.Tensor) -> torch.Tensor:
        if wave.dim() != 2:
            raise ValueError(f"expected [B, samples], got {tuple(wave.shape)}")
        x = wave.to(self.device, torch.float32).contiguous()
        B, N = x.shape
        if N < N_FFT // 2 + 1:
            raise ValueError(f"audio too short for a reflect-padded STFT: {N} samples")
        self.reserve(B, N)
        out = torch.empty(B, OUT_DIM, dtype=torch.float32, device=self.device)
        rc = self.lib.wise_htsat_forward2(self.wb.data_ptr(), self.pf.data_ptr(), x.data_ptr(), B, N, out.data_ptr(),
                                          self._ws.data_ptr(), self._ws.numel(), self._flags, _lib.stream_ptr())
        _lib.check(rc, "wise_htsat_forward")
        self._last = (B, N)
        return out

    def forward_pipelined(self, wave: torch.Tensor):
        """Enqueue one batch of clips and return a handle at once (`.result()` -> embeddings): successive calls
        alternate between two slots, each with its own stream and workspace, so two batches are in flight (same
        scheme and same caveats as VitEngine.forward_pipelined)."""
        from .vit import PendingEmbeddings
        if wave.dim() != 2:
            raise ValueError(f"expected [B, samples], got {tuple(wave.shape)}")
        x = wave.to(self.device, torch.float32).contiguous()
        B, N = x.shape
        if N < N_FFT // 2 + 1:
            raise ValueError(f"audio too short for a reflect-padded STFT: {N} samples")
        if not hasattr(self, "_slots"):
            from .._streams import concurrent_streams   # streams SEEN to run side by side (two on one hardware queue: no overlap)
            self._slots, self._next_slot = [{"stream": st, "ws": None} for st in concurrent_streams(
                max(2, int(getattr(self, "batches_in_flight", 2))), self.device)], 0
        need = self.lib.wise_htsat_workspace_bytes(B, N)
        slot = self._slots[self._next_slot]
        self._next_slot = (self._next_slot + 1) % len(self._slots)
        if slot["ws"] is None or slot["ws"].numel() < need:
            # the slot's previous forward may still be running in the old workspace (allocated on the caller's stream,
            # used on the slot's): wait for it before the allocator may reuse that block
            slot["stream"].synchronize()
            slot["ws"] = torch.empty(need, dtype=torch.uint8, device=self.device)
        slot["stream"].wait_stream(torch.cuda.current_stream(self.device))
        out = torch.empty(B, OUT_DIM, dtype=torch.float32, device=self.device)
        x.record_stream(slot["stream"])
        out.record_stream(slot["stream"])
        self.lib.wise_overlap_hint(1)      # this batch runs beside the other slot's: tile for co-residency
        rc = self.lib.wise_htsat_forward2(self.wb.data_ptr(), self.pf.data_ptr(), x.data_ptr(), B, N, out.data_ptr(),
                                          slot["ws"].data_ptr(), slot["ws"].numel(), self._flags, slot["stream"].cuda_stream)
        self.lib.wise_overlap_hint(0)
        _lib.check(rc, "wise_htsat_forward")
        done = torch.cuda.Event()
        done.record(slot["stream"])
        return PendingEmbeddings(out, done)

    def tap(self, what: int, rows: int, cols: int) -> torch.Tensor:
        """parity taps: 0 = log-mel+bn [B*frames,64] fp32, 1 = residual stream x fp32 [rows, cols]."""
        out = torch.empty(rows, cols, dtype=torch.float32, device=self.device)
        if what == 1 and self.ln_fold:
            what = 2                      # the last stage's rows are a hi + lo stream in fold mode
        _lib.check(self.lib.wise_htsat_tap(what, self._ws.data_ptr(), self._last[0], self._last[1], out.data_ptr(),
                                           rows * cols, _lib.stream_ptr()), "wise_htsat_tap")
        return out


def mlp_stream_weights(w1: torch.Tensor, w2: torch.Tensor) -> torch.Tensor:
    """fc1 [4C, C] and fc2 [C, 4C] -> ONE tensor of 8 C^2 elements in the order wise_mlp_stream consumes them (include/wise_hip.h):
    per step s of 32 hidden units the 2 * C/32 fc1 fragments (j, ks) and then the C/16 fc2 fragments (jn), a fragment being
    [lane = 16 g + l][8]:  fc1: W1[32 s + 16 j + l, 32 ks + 8 g + e];  fc2: W2[16 jn + l, 32 s + 16 (e // 4) + 4 g + e % 4]."""
    F, C = w1.shape
    assert F == 4 * C and w2.shape == (C, F) and C % 32 == 0
    ns = F // 32
    a = w1.reshape(ns, 2, 16, C // 32, 4, 8).permute(0, 1, 3, 4, 2, 5)          # s, j, ks, g, l, e
    b = w2.reshape(C // 16, 16, ns, 2, 4, 4).permute(2, 0, 4, 1, 3, 5)          # s, jn, g, l, half, e4
    return torch.cat([a.reshape(ns, -1), b.reshape(ns, -1)], dim=1).reshape(-1).contiguous()


def swin_qkv_stream(w_qkv: torch.Tensor, b_qkv: torch.Tensor):
    """W_qkv [3C, C], b [3C] -> (the weights as wise_swin_qkv_attn's stream, the bias in its step order): step s = 3 p + t (p: pair of
    heads, t: q / k / v) holds rows t*C + 48 p .. + 47 as fragments (j < 3, ks) of [lane = 16 g + l][8] (include/wise_hip.h)."""
    C3, C = w_qkv.shape
    assert C3 == 3 * C and C % 96 == 0
    npair = C // 48
    w = w_qkv.reshape(3, npair, 3, 16, C // 32, 4, 8).permute(1, 0, 2, 4, 5, 3, 6)      # p, t, j, ks, g, l, e
    b = b_qkv.reshape(3, npair, 48).permute(1, 0, 2)
    return w.reshape(-1).contiguous(), b.reshape(-1).contiguous()


def pack_htsat_weights(sd: Dict[str, torch.Tensor], fold: bool = False, mlp_stream: bool = False, attn_stream: bool = False):
    """state dict -> (bf16 blob, fp32 blob) in the order wise_htsat_layout() documents (CPU tensors).
    fold: the qkv / fc1 weights and biases of stages 2 - 4 with norm1 / norm2 folded in (vit.fold_layernorm: gamma-scaled,
    row-centred weights, bias + W beta) — what wise_htsat_forward2 flags bit 0 expects; the norm slots stay (unread there).
    mlp_stream: the fc1 + fc2 slots of stages 2 and 3 (C = 192, 384) hold the two matrices as wise_mlp_stream's weight stream
    (mlp_stream_weights) — what flags bit 1 expects.

    bf16: per block qkv [3C,C], proj [C,C], fc1 [4C,C], fc2 [C,4C]; per stage<3 reduction [2C,4C];
          then projection.linear1 [1024,768], linear2 [1024,1024]
    fp32: bn0 scale[64], shift[64] (running stats folded), mel filterbank (sparse: start[64], len[64], weights^T [MELW,64]),
          hann[1024], patch conv w [96,16], b[96], norm w,b; per block norm1 w,b, rel-pos bias expanded [heads,64,64],
          qkv bias, proj bias, norm2 w,b, fc1 bias, fc2 bias; per stage<3 merge norm w,b [4C]; final norm w,b;
          projection LN w,b
    """
    from . import htsat_frontend as fe

    f32 = lambda k: sd[k].detach().to(torch.float32).cpu()
    pre = "base.htsat."
    wb, pf = [], []
    scale = f32(pre + "bn0.weight") / torch.sqrt(f32(pre + "bn0.running_var") + 1e-5)
    shift = f32(pre + "bn0.bias") - f32(pre + "bn0.running_mean") * scale
    start, length, weights = fe.sparse_mel()
    assert int(length.max()) <= 16, "HTSAT front end: the kernel applies 16 weights per mel band (csrc/htsat.hip)"
    pf += [scale, shift, start.to(torch.float32), length.to(torch.float32), weights.t().contiguous().reshape(-1),
           fe.hann_periodic(),
           f32(pre + "patch_embed.proj.weight").reshape(-1), f32(pre + "patch_embed.proj.bias"),
           f32(pre + "patch_embed.norm.weight"), f32(pre + "patch_embed.norm.bias")]
    rel_idx = fe.rel_pos_index().reshape(-1)
    for i, depth in enumerate(DEPTHS):
        for j in range(depth):
            p = f"{pre}layers.{i}.blocks.{j}."
            w_qkv, b_qkv = f32(p + "attn.qkv.weight"), f32(p + "attn.qkv.bias")
            w_fc1, b_fc1 = f32(p + "mlp.fc1.weight"), f32(p + "mlp.fc1.bias")
            if fold and i >= 1:
                from .vit import fold_layernorm
                w_qkv, b_qkv = fold_layernorm(w_qkv, b_qkv, f32(p + "norm1.weight"), f32(p + "norm1.bias"))
                w_fc1, b_fc1 = fold_layernorm(w_fc1, b_fc1, f32(p + "norm2.weight"), f32(p + "norm2.bias"))
            if attn_stream and i in (1, 2):      # W_qkv and its bias as wise_swin_qkv_attn's stream (same sizes, another order)
                w_qkv, b_qkv = swin_qkv_stream(w_qkv, b_qkv)
            w_fc2 = f32(p + "mlp.fc2.weight")
            if mlp_stream and i in (1, 2):
                mlp = [mlp_stream_weights(w_fc1, w_fc2)]
            else:
                mlp = [w_fc1.reshape(-1), w_fc2.reshape(-1)]
            wb += [w_qkv.reshape(-1), f32(p + "attn.proj.weight").reshape(-1)] + mlp
            bias = f32(p + "attn.relative_position_bias_table")[rel_idx].reshape(64, 64, HEADS[i]).permute(2, 0, 1)
            pf += [f32(p + "norm1.weight"), f32(p + "norm1.bias"), bias.contiguous().reshape(-1),
                   b_qkv, f32(p + "attn.proj.bias"), f32(p + "norm2.weight"),
                   f32(p + "norm2.bias"), b_fc1, f32(p + "mlp.fc2.bias")]
        if i < 3:
            p = f"{pre}layers.{i}.downsample."
            wb.append(f32(p + "reduction.weight").reshape(-1))
            pf += [f32(p + "norm.weight"), f32(p + "norm.bias")]
    wb += [f32("projection.linear1.weight").reshape(-1), f32("projection.linear2.weight").reshape(-1)]
    pf += [f32(pre + "norm.weight"), f32(pre + "norm.bias"), f32("projection.layer_norm.weight"),
           f32("projection.layer_norm.bias")]
    return torch.cat(wb).to(torch.bfloat16).contiguous(), torch.cat(pf).contiguous()
