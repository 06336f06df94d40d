"""Host side of the MS-CLAP (version 2022) audio encoder — PANNs Cnn14 + msclap Projection: state-dict layout, seeded
initialiser, weight packing (BatchNorm folded into the convolutions), and the engine that drives `wise_cnn14_forward`
(include/wise_hip.h).

Weights are addressed by msclap 1.3.3 state-dict keys under `clap.audio_encoder.` (what `msclap.CLAP(version='2022')`
loads, reference call site src/feature/microsoft_clap.py:31), so a real checkpoint is a pure data problem.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Tuple

import torch

from .. import _lib

N_FFT, HOP, N_MELS = 1024, 320, 64
FMAX = 14000.0                       # msclap config_2022 (the 2023 config stops at 8000)
CHANNELS = (64, 128, 256, 512, 1024, 2048)
EMB = 2048
OUT_DIM = 1024
MIN_FRAMES = 32                      # five 2x2 poolings must leave at least one time step


def state_dict_keys() -> List[Tuple[str, Tuple[int, ...]]]:
    """(key, shape) in the order the seeded initialiser draws them (STFT kernels / mel filterbank are fixed functions of
    the config and are rebuilt, not loaded; fc_audioset feeds only the classification output, which the reference
    drops at microsoft_clap.py:49 `[0]`)."""
    pre = "base."
    keys = [(pre + f"bn0.{k}", (N_MELS,)) for k in ("weight", "bias", "running_mean", "running_var")]
    cin = 1
    for i, cout in enumerate(CHANNELS):
        p = f"{pre}conv_block{i + 1}."
        keys.append((p + "conv1.weight", (cout, cin, 3, 3)))
        keys.append((p + "conv2.weight", (cout, cout, 3, 3)))
        for bn in ("bn1", "bn2"):
            keys += [(p + f"{bn}.{k}", (cout,)) for k in ("weight", "bias", "running_mean", "running_var")]
        cin = cout
    keys += [(pre + "fc1.weight", (EMB, EMB)), (pre + "fc1.bias", (EMB,)),
             ("projection.linear1.weight", (OUT_DIM, EMB)), ("projection.linear2.weight", (OUT_DIM, OUT_DIM)),
             ("projection.layer_norm.weight", (OUT_DIM,)), ("projection.layer_norm.bias", (OUT_DIM,))]
    return keys


def random_cnn14_state_dict(seed: int = 0) -> Dict[str, torch.Tensor]:
    """Seeded fp32 weights (no checkpoint exists offline); one CPU generator, `state_dict_keys` order.  He-style
    convolution weights and BatchNorm statistics near (0, 1) keep the activations O(1) through the twelve layers."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for key, shape in state_dict_keys():
        n = torch.randn(shape, generator=g, dtype=torch.float32)
        if key.endswith("bn0.running_mean"):
            t = -30.0 + 5.0 * n            # log-mel dB of 0.1-amplitude noise sits around -30 dB
        elif key.endswith("bn0.running_var"):
            t = 100.0 * (1.0 + 0.2 * n).abs() + 1.0
        elif key.endswith("running_mean"):
            t = 0.1 * n
        elif key.endswith("running_var"):
            t = (1.0 + 0.2 * n).abs() + 0.1
        elif key.endswith("conv1.weight") or key.endswith("conv2.weight"):
            t = n * (2.0 / (shape[1] * 9)) ** 0.5
        elif key.endswith(".weight") and len(shape) == 2:
            t = n * shape[1] ** -0.5
        elif key.endswith(".weight"):
            t = 1.0 + 0.1 * n              # BatchNorm / LayerNorm scales
        elif key.endswith(".bias"):
            t = 0.1 * n
        else:
            raise KeyError(key)
        sd[key] = t.contiguous()
    return sd


def flops_per_clip(samples: int) -> int:
    """multiply-adds x 2 of the twelve convolutions, fc1 and the projection for one clip of `samples` samples"""
    T, F, cin, total = samples // HOP + 1, N_MELS, 1, 0
    for i, cout in enumerate(CHANNELS):
        total += 2 * T * F * 9 * (cin * cout + cout * cout)
        if i < len(CHANNELS) - 1:
            T, F = T // 2, F // 2
        cin = cout
    return total + 2 * (EMB * EMB + EMB * OUT_DIM + OUT_DIM * OUT_DIM)


def _fold(sd, p: str, conv: str, bn: str):
    """conv weight [Cout, Cin, 3, 3] and its BatchNorm -> ([Cout, 9, Cin] scaled, shift [Cout]); k = (kh*3 + kw)*Cin + c"""
    f32 = lambda k: sd[k].detach().to(torch.float32).cpu()
    scale = f32(p + bn + ".weight") / torch.sqrt(f32(p + bn + ".running_var") + 1e-5)
    shift = f32(p + bn + ".bias") - f32(p + bn + ".running_mean") * scale
    w = f32(p + conv + ".weight") * scale[:, None, None, None]
    return w.permute(0, 2, 3, 1).contiguous().reshape(w.shape[0], -1), shift


def pack_cnn14_weights(sd: Dict[str, torch.Tensor]):
    """state dict -> (bf16 blob, fp32 blob) in the order wise_cnn14_layout() documents (CPU tensors).

    bf16: the eleven MFMA convolutions [Cout, 9*Cin] (block 1 conv 2, then both of blocks 2-6), BatchNorm scale folded;
          fc1 [2048, 2048]; projection.linear1 [1024, 2048], linear2 [1024, 1024]
    fp32: bn0 scale[64], shift[64] (running stats folded), mel filterbank (sparse: start[64], len[64], weights^T [MELW,64]),
          hann[1024], block 1 conv 1 [64, 9] (scaled) and its shift [64]; the eleven BatchNorm shifts; fc1 bias;
          projection LN w, b
    """
    from . import htsat_frontend as fe

    f32 = lambda k: sd[k].detach().to(torch.float32).cpu()
    pre = "base."
    scale = f32(pre + "bn0.weight") / torch.sqrt(f32(pre + "bn0.running_var") + 1e-5)
    shift = f32(pre + "bn0.bias") - f32(pre + "bn0.running_mean") * scale
    start, length, weights = fe.sparse_mel(FMAX)
    w0, s0 = _fold(sd, pre + "conv_block1.", "conv1", "bn1")
    pf = [scale, shift, start.to(torch.float32), length.to(torch.float32), weights.t().contiguous().reshape(-1),
          fe.hann_periodic(), w0.reshape(-1), s0]
    wb = []
    for i in range(len(CHANNELS)):
        p = f"{pre}conv_block{i + 1}."
        for j, (conv, bn) in enumerate((("conv1", "bn1"), ("conv2", "bn2"))):
            if i == 0 and j == 0:
                continue
            w, s = _fold(sd, p, conv, bn)
            wb.append(w.reshape(-1))
            pf.append(s)
    wb += [f32(pre + "fc1.weight").reshape(-1), f32("projection.linear1.weight").reshape(-1),
           f32("projection.linear2.weight").reshape(-1)]
    pf += [f32(pre + "fc1.bias"), f32("projection.layer_norm.weight"), f32("projection.layer_norm.bias")]
    return torch.cat(wb).to(torch.bfloat16).contiguous(), torch.cat(pf).contiguous()


class Cnn14Engine:
    """Device copies of the packed weights + workspace; forward(wave [B,N] fp32) -> [B,1024] fp32 device tensor,
    L2-normalised (microsoft_clap.py:49-50).  Same interface as HtsatEngine."""

    def __init__(self, sd: Dict[str, torch.Tensor], device: str = "cuda", max_batch: int = 8,
                 max_samples: int = 480000):
        self.lib = _lib.lib()
        self.device = torch.device(device)
        nb, nf = C.c_int64(), C.c_int64()
        _lib.check(self.lib.wise_cnn14_layout(C.byref(nb), C.byref(nf)), "wise_cnn14_layout")
        wb, pf = pack_cnn14_weights(sd)
        if wb.numel() != nb.value or pf.numel() != nf.value:
            raise RuntimeError(f"Cnn14 blob size mismatch: packed {wb.numel()}/{pf.numel()}, "
                               f"library expects {nb.value}/{nf.value}")
        self.wb, self.pf = wb.to(self.device), pf.to(self.device)
        self._ws = None
        self._ws_bytes = 0
        self._last = (0, 0)
        self.reserve(max_batch, max_samples)

    def reserve(self, batch: int, samples: int):
        n = self.lib.wise_cnn14_workspace_bytes(batch, samples)
        if n == 0:
            raise ValueError(f"Cnn14: batch {batch} x {samples} samples unsupported "
                             f"(at least {MIN_FRAMES} STFT frames = {(MIN_FRAMES - 1) * HOP} samples)")
        if n > self._ws_bytes:
            self._ws = torch.empty(n, dtype=torch.uint8, device=self.device)
            self._ws_bytes = n

    def forward(self, wave: torch.Tensor) -> torch.Tensor:
        if wave.dim() != 2:
            raise ValueError(f"expected [B, samples], got {tuple(wave.shape)}")
        x = wave.to(self.device, torch.float32).contiguous()
        B, N = x.shape
        self.reserve(B, N)
        out = torch.empty(B, OUT_DIM, dtype=torch.float32, device=self.device)
        rc = self.lib.wise_cnn14_forward(self.wb.data_ptr(), self.pf.data_ptr(), x.data_ptr(), B, N, out.data_ptr(),
                                         self._ws.data_ptr(), self._ws.numel(), _lib.stream_ptr())
        _lib.check(rc, "wise_cnn14_forward")
        self._last = (B, N)
        return out

    def forward_pipelined(self, wave: torch.Tensor):
        """Enqueue one batch of clips and return a handle at once (`.result()` -> embeddings): successive calls
        alternate between two slots, each with its own stream and workspace (the scheme of HtsatEngine)."""
        from .vit import PendingEmbeddings
        if wave.dim() != 2:
            raise ValueError(f"expected [B, samples], got {tuple(wave.shape)}")
        x = wave.to(self.device, torch.float32).contiguous()
        B, N = x.shape
        need = self.lib.wise_cnn14_workspace_bytes(B, N)
        if need == 0:
            raise ValueError(f"Cnn14: batch {B} x {N} samples unsupported")
        if not hasattr(self, "_slots"):
            from .._streams import concurrent_streams   # streams SEEN to run side by side (two on one hardware queue: no overlap)
            self._slots, self._next_slot = [{"stream": st, "ws": None} for st in concurrent_streams(2, self.device)], 0
        slot = self._slots[self._next_slot]
        self._next_slot ^= 1
        if slot["ws"] is None or slot["ws"].numel() < need:
            slot["stream"].synchronize()
            slot["ws"] = torch.empty(need, dtype=torch.uint8, device=self.device)
        slot["stream"].wait_stream(torch.cuda.current_stream(self.device))
        out = torch.empty(B, OUT_DIM, dtype=torch.float32, device=self.device)
        x.record_stream(slot["stream"])
        out.record_stream(slot["stream"])
        self.lib.wise_overlap_hint(1)
        rc = self.lib.wise_cnn14_forward(self.wb.data_ptr(), self.pf.data_ptr(), x.data_ptr(), B, N, out.data_ptr(),
                                         slot["ws"].data_ptr(), slot["ws"].numel(), slot["stream"].cuda_stream)
        self.lib.wise_overlap_hint(0)
        _lib.check(rc, "wise_cnn14_forward")
        done = torch.cuda.Event()
        done.record(slot["stream"])
        return PendingEmbeddings(out, done)

    def tap(self, what: int) -> torch.Tensor:
        """parity taps of the last forward: 0 = log-mel+bn fp32 [B, frames, 64], 1 = pooled latent bf16 [B, 2048],
        2 = fc1 output bf16 [B, 2048]."""
        B, N = self._last
        T = N // HOP + 1
        if what == 0:
            out = torch.empty(B, T, N_MELS, dtype=torch.float32, device=self.device)
        else:
            out = torch.empty(B, EMB, dtype=torch.bfloat16, device=self.device)
        _lib.check(self.lib.wise_cnn14_tap(what, self._ws.data_ptr(), B, N, out.data_ptr(),
                                           out.numel() * out.element_size(), _lib.stream_ptr()), "wise_cnn14_tap")
        return out


def conv3x3_relu(x: torch.Tensor, wt: torch.Tensor, bias: torch.Tensor, pool: bool = False) -> torch.Tensor:
    """relu(conv3x3(x) + bias), optionally followed by the fused 2x2 average pooling, through `wise_conv3x3_relu_bf16`:
    x [B, T, F, Cin] bf16 (position-major), wt [Cout, 9*Cin] bf16, bias [Cout] fp32 -> [B, T, F, Cout] bf16 (or
    [B, T//2, F//2, Cout]).  The building block of the engine, for parity tests."""
    B, T, F, Cin = x.shape
    Cout = wt.shape[0]
    To, Fo = (T // 2, F // 2) if pool else (T, F)
    rows = (B * To * Fo + 255) // 256 * 256
    out = torch.empty(rows, Cout, dtype=torch.bfloat16, device=x.device)
    zeros = torch.zeros(64, dtype=torch.bfloat16, device=x.device)
    rc = _lib.lib().wise_conv3x3_relu_bf16(x.contiguous().data_ptr(), wt.contiguous().data_ptr(),
                                          bias.contiguous().data_ptr(), zeros.data_ptr(), B, T, F, Cin, Cout, int(pool),
                                          out.data_ptr(), _lib.stream_ptr())
    _lib.check(rc, "wise_conv3x3_relu_bf16")
    return out[: B * To * Fo].reshape(B, To, Fo, Cout)
