"""Fixed (non-learned) tables of the HTSAT front-end and window attention, built on the host and
shipped to the GPU inside the fp32 parameter blob: periodic Hann window, the librosa-style Slaney mel
filterbank of msclap's 2023 config (sr 44100, n_fft 1024, 64 bands, 50..8000 Hz) in sparse form, and
Swin's relative-position index for an 8x8 window."""
from __future__ import annotations

import math

import numpy as np
import torch

N_FFT, N_MELS, SR, FMIN, FMAX = 1024, 64, 44100, 50.0, 8000.0
MELW = 36  # max non-zero FFT bins per band kept in the sparse table (asserted below; csrc/transformer.h FRONT_MELW)


def hann_periodic(n: int = N_FFT) -> torch.Tensor:
    k = torch.arange(n, dtype=torch.float64)
    return (0.5 - 0.5 * torch.cos(2 * math.pi * k / n)).to(torch.float32)


def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    logstep = np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_hz / f_sp + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, f / f_sp)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filterbank(fmax: float = FMAX) -> np.ndarray:
    """[64, 513] float32: triangular filters on the Slaney mel scale, area-normalised (librosa.filters.mel).
    fmax: 8000 in msclap's 2023 config (HTSAT), 14000 in the 2022 config (Cnn14)."""
    fftfreqs = np.linspace(0, SR / 2.0, 1 + N_FFT // 2)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(FMIN), _hz_to_mel(fmax), N_MELS + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    w = np.zeros((N_MELS, 1 + N_FFT // 2))
    for i in range(N_MELS):
        w[i] = np.maximum(0, np.minimum(-ramps[i] / fdiff[i], ramps[i + 2] / fdiff[i + 1]))
    enorm = 2.0 / (mel_f[2: N_MELS + 2] - mel_f[:N_MELS])
    return (w * enorm[:, None]).astype(np.float32)


def sparse_mel(fmax: float = FMAX):
    """-> (start int32[64], length int32[64], weights float32[64, MELW]) with weights[b, j] = fb[b, start[b]+j]."""
    fb = mel_filterbank(fmax)
    start = np.zeros(N_MELS, np.int32)
    length = np.zeros(N_MELS, np.int32)
    wts = np.zeros((N_MELS, MELW), np.float32)
    for b in range(N_MELS):
        nz = np.nonzero(fb[b])[0]
        if nz.size:
            start[b], length[b] = nz[0], nz[-1] - nz[0] + 1
            assert length[b] <= MELW, f"band {b} spans {length[b]} bins > MELW"
            wts[b, : length[b]] = fb[b, nz[0]: nz[-1] + 1]
    return torch.from_numpy(start), torch.from_numpy(length), torch.from_numpy(wts)


def rel_pos_index(ws: int = 8) -> torch.Tensor:
    coords = torch.stack(torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")).flatten(1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)  # [query, key]
