"""ORACLE (test infrastructure — never imported by the product path under wise_amd/).

CPU fp32 restatement of the audio path the reference runs at
/root/reference/src/feature/microsoft_clap.py:45-51:

    audio_embeddings = self.model.clap.audio_encoder(preprocessed_audio)[0]      (:49)
    audio_embeddings = audio_embeddings / torch.norm(audio_embeddings, dim=-1, keepdim=True)   (:50)

`audio_encoder` lives in the un-vendored dependency msclap==1.3.3 (/root/reference/requirements.txt:25-26):
AudioEncoder = HTSAT_Swin_Transformer (with its torchlibrosa front-end INSIDE the model) + Projection.
Restated here from the published architecture, version '2023' config:
  front-end  STFT n_fft 1024 / hop 320 / periodic hann / center, reflect pad -> power -> 64 log-mel bands
             (librosa Slaney filterbank, sr 44100, fmin 50, fmax 8000) -> 10*log10(max(.,1e-10))
             WISE feeds 48 kHz samples without resampling (extract-features.py:292); the model treats them as 44.1 kHz.
  bn0        BatchNorm2d over the 64 mel bins (eval mode)
  image      time axis bicubic (align_corners=True) to 1024 frames, folded to 256x256 (freq_ratio 4)
  body       patch embed conv 4x4/4 (1->96) + LN; 4 Swin stages, depths 2-2-6-2, dims 96..768, heads 4..32,
             window 8, shift 4 on odd blocks (none when the map is one window), rel-pos bias, mask -100;
             PatchMerging (x0|x1|x2|x3 -> LN(4C) -> Linear 4C->2C, no bias); final LN; mean over tokens -> 768
  projection linear1 (768->1024, no bias), GELU, linear2 (1024->1024, no bias), LayerNorm(e1 + e2)

PINNING: msclap is not installed and no checkpoint exists offline.  The Swin body (bn0 ... pooled 768-d
latent) is pinned against transformers' ClapAudioModel — an independent implementation of the same
HTSAT lineage that IS in the container — on the same seeded weights (oracle/make_golden_htsat.py).  The
STFT is pinned against torch.stft.  The mel filterbank restates librosa.filters.mel (Slaney scale + norm)
and the Projection restates msclap's; both, and the front-end constants (fmax, sr), are UNPINNED offline.
Inputs longer than 1024 frames (6.8 s at hop 320): the first 1024 frames are used (msclap's own behaviour
for that case could not be checked offline).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import numpy as np
import torch

N_FFT = 1024
HOP = 320
N_MELS = 64
SR = 44100
FMIN = 50.0
FMAX = 8000.0
SPEC_SIZE = 256
FREQ_RATIO = 4
TARGET_T = SPEC_SIZE * FREQ_RATIO  # 1024
WINDOW = 8
DEPTHS = (2, 2, 6, 2)
HEADS = (4, 8, 16, 32)
EMBED = 96
OUT_DIM = 1024


# ---------------------------------------------------------------------------- front-end
def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, mels)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    freqs = f_sp * m
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), freqs)


def mel_filterbank(sr: int = SR, n_fft: int = N_FFT, n_mels: int = N_MELS, fmin: float = FMIN,
                   fmax: float = FMAX) -> np.ndarray:
    """librosa.filters.mel(htk=False, norm='slaney') -> [n_mels, 1 + n_fft//2] float32."""
    fftfreqs = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    w = np.zeros((n_mels, 1 + n_fft // 2))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2: n_mels + 2] - mel_f[:n_mels])
    return (w * enorm[:, None]).astype(np.float32)


def hann_periodic(n: int = N_FFT) -> torch.Tensor:
    k = torch.arange(n, dtype=torch.float64)
    return (0.5 - 0.5 * torch.cos(2 * math.pi * k / n)).to(torch.float32)


def power_spectrogram(wave: torch.Tensor) -> torch.Tensor:
    """[B, N] -> [B, frames, 513] |STFT|^2 (torchlibrosa Spectrogram: center, reflect, power 2); explicit DFT."""
    B, N = wave.shape
    x = torch.nn.functional.pad(wave.unsqueeze(1), (N_FFT // 2, N_FFT // 2), mode="reflect").squeeze(1)
    frames = 1 + N // HOP
    idx = torch.arange(frames).unsqueeze(1) * HOP + torch.arange(N_FFT).unsqueeze(0)
    seg = x[:, idx] * hann_periodic()  # [B, frames, 1024]
    k = torch.arange(N_FFT // 2 + 1, dtype=torch.float64)
    n = torch.arange(N_FFT, dtype=torch.float64)
    ang = 2 * math.pi * torch.outer(n, k) / N_FFT
    cr, ci = torch.cos(ang).to(torch.float32), -torch.sin(ang).to(torch.float32)
    re, im = seg @ cr, seg @ ci
    return re * re + im * im


def logmel(wave: torch.Tensor) -> torch.Tensor:
    """[B,N] -> [B, frames, 64] log-mel in dB (ref 1.0, amin 1e-10, no top_db)."""
    p = power_spectrogram(wave)
    mel = p @ torch.from_numpy(mel_filterbank()).t()
    return 10.0 * torch.log10(torch.clamp(mel, min=1e-10))


# ---------------------------------------------------------------------------- body
def layer_norm(x, w, b, eps=1e-5):
    mean = x.mean(dim=-1, keepdim=True)
    var = ((x - mean) ** 2).mean(dim=-1, keepdim=True)
    return (x - mean) / torch.sqrt(var + eps) * w + b


def gelu(x):
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def bicubic_time(x: torch.Tensor, t_out: int) -> torch.Tensor:
    """[B, T, F] -> [B, t_out, F]; torch upsample_bicubic2d semantics (A=-0.75, align_corners=True, clamped taps)."""
    B, T, F = x.shape
    if T == t_out:
        return x
    A = -0.75
    scale = (T - 1) / (t_out - 1) if t_out > 1 else 0.0
    src = torch.arange(t_out, dtype=torch.float32) * scale
    i0 = torch.floor(src).to(torch.int64)
    t = (src - i0.to(torch.float32))

    def c1(v):  # |v| <= 1
        return ((A + 2) * v - (A + 3)) * v * v + 1

    def c2(v):  # 1 < |v| < 2
        return ((A * v - 5 * A) * v + 8 * A) * v - 4 * A

    w = torch.stack([c2(t + 1), c1(t), c1(1 - t), c2(2 - t)], dim=1)  # [t_out, 4]
    out = torch.zeros(B, t_out, F, dtype=x.dtype)
    for j in range(4):
        idx = torch.clamp(i0 - 1 + j, 0, T - 1)
        out += x[:, idx, :] * w[:, j].view(1, t_out, 1)
    return out


def mel_to_image(melbn: torch.Tensor) -> torch.Tensor:
    """[B, T<=1024, 64] -> [B, 256, 256]: image[r*64 + f][t'] = mel[t = r*256 + t'][f]."""
    B = melbn.shape[0]
    x = bicubic_time(melbn[:, :TARGET_T, :], TARGET_T)  # [B,1024,64]
    x = x.reshape(B, FREQ_RATIO, TARGET_T // FREQ_RATIO, N_MELS)  # [B, r, t', f]
    return x.permute(0, 1, 3, 2).reshape(B, FREQ_RATIO * N_MELS, TARGET_T // FREQ_RATIO)


def rel_pos_index(ws: int = WINDOW) -> torch.Tensor:
    coords = torch.stack(torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")).flatten(1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)  # [ws*ws (query), ws*ws (key)]


def shift_mask(H: int, W: int, ws: int, shift: int) -> torch.Tensor:
    """[nWindows, ws*ws, ws*ws] additive mask (0 / -100) of the cyclic-shift regions."""
    img = torch.zeros(H, W)
    cnt = 0
    for hs in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
        for wsl in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            img[hs, wsl] = cnt
            cnt += 1
    mw = img.reshape(H // ws, ws, W // ws, ws).permute(0, 2, 1, 3).reshape(-1, ws * ws)
    m = mw.unsqueeze(1) - mw.unsqueeze(2)
    return torch.where(m != 0, torch.tensor(-100.0), torch.tensor(0.0))


def swin_block(x, p, sd, H, W, heads, shift):
    B, T, Cd = x.shape
    ws = WINDOW
    dh = Cd // heads
    h = layer_norm(x, sd[p + "norm1.weight"], sd[p + "norm1.bias"]).reshape(B, H, W, Cd)
    if shift > 0:
        h = torch.roll(h, shifts=(-shift, -shift), dims=(1, 2))
    win = h.reshape(B, H // ws, ws, W // ws, ws, Cd).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws, Cd)
    qkv = win @ sd[p + "attn.qkv.weight"].t() + sd[p + "attn.qkv.bias"]
    nW = win.shape[0]
    qkv = qkv.reshape(nW, ws * ws, 3, heads, dh).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    s = (q * (dh ** -0.5)) @ k.transpose(-1, -2)
    bias = sd[p + "attn.relative_position_bias_table"][rel_pos_index().reshape(-1)].reshape(ws * ws, ws * ws, heads)
    s = s + bias.permute(2, 0, 1).unsqueeze(0)
    if shift > 0:
        m = shift_mask(H, W, ws, shift)
        s = s.reshape(B, m.shape[0], heads, ws * ws, ws * ws) + m.unsqueeze(1).unsqueeze(0)
        s = s.reshape(nW, heads, ws * ws, ws * ws)
    s = s - s.max(dim=-1, keepdim=True).values
    e = torch.exp(s)
    pr = e / e.sum(dim=-1, keepdim=True)
    o = (pr @ v).transpose(1, 2).reshape(nW, ws * ws, Cd)
    o = o @ sd[p + "attn.proj.weight"].t() + sd[p + "attn.proj.bias"]
    o = o.reshape(B, H // ws, W // ws, ws, ws, Cd).permute(0, 1, 3, 2, 4, 5).reshape(B, H, W, Cd)
    if shift > 0:
        o = torch.roll(o, shifts=(shift, shift), dims=(1, 2))
    x = x + o.reshape(B, T, Cd)
    h = layer_norm(x, sd[p + "norm2.weight"], sd[p + "norm2.bias"])
    h = gelu(h @ sd[p + "mlp.fc1.weight"].t() + sd[p + "mlp.fc1.bias"])
    return x + h @ sd[p + "mlp.fc2.weight"].t() + sd[p + "mlp.fc2.bias"]


def patch_merge(x, p, sd, H, W):
    B, T, Cd = x.shape
    g = x.reshape(B, H, W, Cd)
    m = torch.cat([g[:, 0::2, 0::2], g[:, 1::2, 0::2], g[:, 0::2, 1::2], g[:, 1::2, 1::2]], dim=-1)
    m = m.reshape(B, -1, 4 * Cd)
    m = layer_norm(m, sd[p + "norm.weight"], sd[p + "norm.bias"])
    return m @ sd[p + "reduction.weight"].t()


def body_forward(sd: Dict[str, torch.Tensor], mel_db: torch.Tensor, taps: Optional[List[torch.Tensor]] = None,
                 stages: int = 4) -> torch.Tensor:
    """log-mel [B, T, 64] -> pooled latent [B, 768] (msclap out_dict['latent_output'])."""
    pre = "base.htsat."
    rm, rv = sd[pre + "bn0.running_mean"], sd[pre + "bn0.running_var"]
    x = (mel_db - rm) / torch.sqrt(rv + 1e-5) * sd[pre + "bn0.weight"] + sd[pre + "bn0.bias"]
    img = mel_to_image(x)  # [B,256,256]
    B = img.shape[0]
    g = SPEC_SIZE // 4
    pw = sd[pre + "patch_embed.proj.weight"].reshape(EMBED, 16)
    patches = img.reshape(B, g, 4, g, 4).permute(0, 1, 3, 2, 4).reshape(B, g * g, 16)
    x = patches @ pw.t() + sd[pre + "patch_embed.proj.bias"]
    x = layer_norm(x, sd[pre + "patch_embed.norm.weight"], sd[pre + "patch_embed.norm.bias"])
    if taps is not None:
        taps.append(x.clone())
    H = W = g
    for i in range(min(stages, 4)):
        for j in range(DEPTHS[i]):
            shift = 0 if (j % 2 == 0 or min(H, W) <= WINDOW) else WINDOW // 2
            x = swin_block(x, f"{pre}layers.{i}.blocks.{j}.", sd, H, W, HEADS[i], shift)
        if taps is not None:
            taps.append(x.clone())
        if i < 3:
            x = patch_merge(x, f"{pre}layers.{i}.downsample.", sd, H, W)
            H, W = H // 2, W // 2
    x = layer_norm(x, sd[pre + "norm.weight"], sd[pre + "norm.bias"])
    return x.mean(dim=1)


def projection(sd, latent):
    e1 = latent @ sd["projection.linear1.weight"].t()
    e2 = gelu(e1) @ sd["projection.linear2.weight"].t()
    return layer_norm(e1 + e2, sd["projection.layer_norm.weight"], sd["projection.layer_norm.bias"])


def htsat_forward(sd: Dict[str, torch.Tensor], wave: torch.Tensor, normalize: bool = True,
                  taps: Optional[List[torch.Tensor]] = None) -> torch.Tensor:
    """wave [B, N] fp32 -> [B, 1024] (audio_encoder(x)[0], then microsoft_clap.py:50's L2 normalise)."""
    out = projection(sd, body_forward(sd, logmel(wave.to(torch.float32)), taps=taps))
    if normalize:
        out = out / torch.norm(out, dim=-1, keepdim=True)
    return out
