"""ORACLE (test infrastructure — never imported by the product path under wise_amd/).

CPU fp32 restatement of the audio path of MS-CLAP version '2022' — what the reference runs at
/root/reference/src/feature/microsoft_clap.py:45-51 when the feature id is `microsoft/clap/2022/...`
(:20-31 accepts every key of msclap's `CLAP.model_name`):

    audio_embeddings = self.model.clap.audio_encoder(preprocessed_audio)[0]      (:49)
    audio_embeddings = audio_embeddings / torch.norm(audio_embeddings, dim=-1, keepdim=True)   (:50)

`audio_encoder` lives in the un-vendored dependency msclap==1.3.3 (/root/reference/requirements.txt:25-26).
For '2022' it is AudioEncoder('Cnn14', d_in 2048, d_out 1024): the PANNs Cnn14 network (torchlibrosa front end
INSIDE the model) followed by msclap's Projection.  Restated here from the published architecture:
  front end  STFT n_fft 1024 / hop 320 / periodic hann / center, reflect pad -> power -> 64 log-mel bands
             (librosa Slaney filterbank, sr 44100, fmin 50, fmax 14000) -> 10*log10(max(., 1e-10))
  bn0        BatchNorm2d over the 64 mel bins (eval mode)
  body       six ConvBlocks (1->64->128->256->512->1024->2048 channels): conv 3x3 pad 1 no bias, BatchNorm2d, ReLU,
             twice; then avg_pool2d 2x2 (floor) after blocks 1-5, nothing after block 6; dropout is identity in eval
  pooling    mean over the mel axis, then max over time + mean over time -> 2048
  fc1        Linear(2048, 2048) + ReLU -> the 'embedding' msclap hands to the projection
  projection linear1 (2048->1024, no bias), GELU, linear2 (1024->1024, no bias), LayerNorm(e1 + e2)

PINNING: msclap is not installed and no checkpoint exists offline; transformers holds no Cnn14.  The body is assembled
from torch's own conv2d / batch_norm / avg_pool2d, the front end is oracle/htsat_ref.py's (STFT pinned against
torch.stft) with fmax 14000; the ASSEMBLY (layer order, pooling, fmax, the fc1 + ReLU in front of the projection) is
**parity unpinned** against msclap — it rests on the published PANNs / msclap sources as recalled in SURVEY.md App. A.2.
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

from . import htsat_ref

FMAX_2022 = 14000.0
CHANNELS = (64, 128, 256, 512, 1024, 2048)
EMB = 2048
OUT_DIM = 1024


def logmel_2022(wave: torch.Tensor) -> torch.Tensor:
    """[B, N] -> [B, frames, 64] log-mel in dB with the 2022 config's filterbank (fmax 14000)."""
    p = htsat_ref.power_spectrogram(wave)
    mel = p @ torch.from_numpy(htsat_ref.mel_filterbank(fmax=FMAX_2022)).t()
    return 10.0 * torch.log10(torch.clamp(mel, min=1e-10))


def _bn(x, sd, pre):
    return F.batch_norm(x, sd[pre + "running_mean"], sd[pre + "running_var"], sd[pre + "weight"], sd[pre + "bias"],
                        training=False, eps=1e-5)


def cnn14_embedding(sd: Dict[str, torch.Tensor], wave: torch.Tensor, taps: Dict[str, torch.Tensor] | None = None):
    """[B, N] fp32 -> [B, 2048] (the 'embedding' entry of Cnn14's output dict)."""
    x = logmel_2022(wave.float()).unsqueeze(1)            # [B, 1, T, 64]
    x = _bn(x.transpose(1, 3), sd, "base.bn0.").transpose(1, 3)
    if taps is not None:
        taps["melbn"] = x[:, 0].clone()
    for i in range(6):
        p = f"base.conv_block{i + 1}."
        x = F.relu(_bn(F.conv2d(x, sd[p + "conv1.weight"], padding=1), sd, p + "bn1."))
        x = F.relu(_bn(F.conv2d(x, sd[p + "conv2.weight"], padding=1), sd, p + "bn2."))
        if i < 5:
            x = F.avg_pool2d(x, kernel_size=2)
        if taps is not None:
            taps[f"block{i + 1}"] = x.clone()                 # [B, C, T', F']
    x = x.mean(dim=3)
    x = x.max(dim=2).values + x.mean(dim=2)
    if taps is not None:
        taps["lat"] = x.clone()
    x = F.relu(F.linear(x, sd["base.fc1.weight"], sd["base.fc1.bias"]))
    if taps is not None:
        taps["emb"] = x.clone()
    return x


def projection(sd: Dict[str, torch.Tensor], x: torch.Tensor) -> torch.Tensor:
    e1 = F.linear(x, sd["projection.linear1.weight"])
    e2 = F.linear(htsat_ref.gelu(e1), sd["projection.linear2.weight"])
    return htsat_ref.layer_norm(e1 + e2, sd["projection.layer_norm.weight"], sd["projection.layer_norm.bias"])


def audio_encoder_2022(sd: Dict[str, torch.Tensor], wave: torch.Tensor, taps=None) -> np.ndarray:
    """the reference's extract_audio_features for version '2022': [B, N] -> [B, 1024] unit rows"""
    with torch.no_grad():
        e = projection(sd, cnn14_embedding(sd, wave, taps))
        return (e / e.norm(dim=-1, keepdim=True)).numpy()
