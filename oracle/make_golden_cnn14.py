"""ORACLE tooling: write tests/golden/cnn14.npz from oracle/cnn14_ref.py on seeded weights and seeded clips.
Authoring container only:
    python -m oracle.make_golden_cnn14
The body has no independent implementation offline to pin against (see the header of cnn14_ref.py); what IS checked
here: the 2022 filterbank restated in the product's packer equals the oracle's.
"""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from oracle import cnn14_ref, htsat_ref  # noqa: E402
from wise_amd.feature import htsat_frontend as fe  # noqa: E402
from wise_amd.feature.cnn14 import FMAX, random_cnn14_state_dict  # noqa: E402

GOLD = ROOT / "tests" / "golden"


def golden_clips():
    """Two 4-s clips at 48 kHz (the reference's segment) and a short one whose sizes are odd at every pooling.  White
    noise alone pools to nearly the same embedding whatever the seed, so the clips differ in CONTENT: noise, a chirp
    with tone bursts, an amplitude-modulated tone."""
    rng = np.random.default_rng(14)
    t = np.arange(192000) / 48000.0
    noise = 0.1 * rng.standard_normal(192000)
    chirp = 0.3 * np.sin(2 * np.pi * (200.0 * t + 1500.0 * t * t)) * (np.sin(2 * np.pi * 3.0 * t) > 0) \
        + 0.2 * np.sin(2 * np.pi * 5200.0 * t) * (np.sin(2 * np.pi * 1.3 * t) > 0.5) + 0.002 * rng.standard_normal(192000)
    w4 = torch.from_numpy(np.stack([noise, chirp]).astype(np.float32))
    t1 = np.arange(33003) / 48000.0
    w1 = (0.25 * np.sin(2 * np.pi * 880.0 * t1) * (0.5 + 0.5 * np.sin(2 * np.pi * 7.0 * t1))
          + 0.01 * rng.standard_normal(33003)).astype(np.float32)[None]
    return w4, torch.from_numpy(w1)


def main():
    torch.set_num_threads(8)
    sd = random_cnn14_state_dict(0)
    assert np.array_equal(fe.mel_filterbank(FMAX), htsat_ref.mel_filterbank(fmax=cnn14_ref.FMAX_2022))
    w4, w1 = golden_clips()
    taps = {}
    out = cnn14_ref.audio_encoder_2022(sd, w4, taps)
    taps1 = {}
    out1 = cnn14_ref.audio_encoder_2022(sd, w1, taps1)
    for k in ("block1", "block3", "block6", "lat", "emb"):
        v = taps[k]
        print(k, tuple(v.shape), "mean |x|", float(v.abs().mean()), "max", float(v.abs().max()),
              "zeros", float((v == 0).float().mean()))
    np.savez_compressed(GOLD / "cnn14.npz", out=out, lat=taps["lat"].numpy(), emb=taps["emb"].numpy(),
                        mel_head=taps["melbn"][:, :8].numpy(), out1=out1, lat1=taps1["lat"].numpy())
    print("wrote", GOLD / "cnn14.npz", out.shape, out1.shape, "cos(out0,out1)", float((out[0] * out[1]).sum()))


if __name__ == "__main__":
    main()
