"""ORACLE tooling: pin oracle/htsat_ref.py's Swin body against transformers' ClapAudioModel, the STFT
against torch.stft, and write tests/golden/htsat.npz.  Authoring container only:
    python -m oracle.make_golden_htsat
"""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from oracle import htsat_ref  # noqa: E402
from wise_amd.feature.htsat import DEPTHS, EMBED, HEADS, checkpoint_like_htsat_state_dict, random_htsat_state_dict  # noqa: E402

GOLD = ROOT / "tests" / "golden"


def hf_audio_model(sd):
    from transformers import ClapAudioConfig, ClapAudioModel

    cfg = ClapAudioConfig(enable_fusion=False, projection_dim=1024, hidden_size=768, window_size=8, num_mel_bins=64,
                          spec_size=256, patch_size=4, patch_stride=[4, 4], depths=list(DEPTHS),
                          num_attention_heads=list(HEADS), patch_embeds_hidden_size=EMBED, drop_path_rate=0.0,
                          attention_probs_dropout_prob=0.0, hidden_dropout_prob=0.0, layer_norm_eps=1e-5)
    m = ClapAudioModel(cfg).eval()
    pre = "base.htsat."
    new = {}
    enc = "audio_encoder."
    for k in ("weight", "bias", "running_mean", "running_var"):
        new[enc + "batch_norm." + k] = sd[pre + "bn0." + k]
    new[enc + "patch_embed.proj.weight"] = sd[pre + "patch_embed.proj.weight"]
    new[enc + "patch_embed.proj.bias"] = sd[pre + "patch_embed.proj.bias"]
    new[enc + "patch_embed.norm.weight"] = sd[pre + "patch_embed.norm.weight"]
    new[enc + "patch_embed.norm.bias"] = sd[pre + "patch_embed.norm.bias"]
    for i, depth in enumerate(DEPTHS):
        Cd = EMBED << i
        for j in range(depth):
            p = f"{pre}layers.{i}.blocks.{j}."
            h = f"{enc}layers.{i}.blocks.{j}."
            new[h + "layernorm_before.weight"] = sd[p + "norm1.weight"]
            new[h + "layernorm_before.bias"] = sd[p + "norm1.bias"]
            new[h + "attention.self.relative_position_bias_table"] = sd[p + "attn.relative_position_bias_table"]
            wq, wk, wv = sd[p + "attn.qkv.weight"].split(Cd, dim=0)
            bq, bk, bv = sd[p + "attn.qkv.bias"].split(Cd, dim=0)
            for n, w_, b_ in (("query", wq, bq), ("key", wk, bk), ("value", wv, bv)):
                new[h + f"attention.self.{n}.weight"] = w_
                new[h + f"attention.self.{n}.bias"] = b_
            new[h + "attention.output.dense.weight"] = sd[p + "attn.proj.weight"]
            new[h + "attention.output.dense.bias"] = sd[p + "attn.proj.bias"]
            new[h + "layernorm_after.weight"] = sd[p + "norm2.weight"]
            new[h + "layernorm_after.bias"] = sd[p + "norm2.bias"]
            new[h + "intermediate.dense.weight"] = sd[p + "mlp.fc1.weight"]
            new[h + "intermediate.dense.bias"] = sd[p + "mlp.fc1.bias"]
            new[h + "output.dense.weight"] = sd[p + "mlp.fc2.weight"]
            new[h + "output.dense.bias"] = sd[p + "mlp.fc2.bias"]
        if i < 3:
            p = f"{pre}layers.{i}.downsample."
            h = f"{enc}layers.{i}.downsample."
            new[h + "norm.weight"] = sd[p + "norm.weight"]
            new[h + "norm.bias"] = sd[p + "norm.bias"]
            new[h + "reduction.weight"] = sd[p + "reduction.weight"]
    new[enc + "norm.weight"] = sd[pre + "norm.weight"]
    new[enc + "norm.bias"] = sd[pre + "norm.bias"]
    missing, unexpected = m.load_state_dict(new, strict=False)
    missing = [k for k in missing if "relative_position_index" not in k and "num_batches_tracked" not in k]
    assert not missing and not unexpected, (missing, unexpected)
    return m


def stress():
    """checkpoint-like statistics (wise_amd/feature/htsat.py::checkpoint_like_htsat_state_dict): pin the oracle's body to
    transformers' ClapAudioModel on these weights again, then write tests/golden/htsat_stress.npz"""
    sd = checkpoint_like_htsat_state_dict(5)
    rng = np.random.default_rng(14)
    wave = torch.from_numpy((0.1 * rng.standard_normal((2, 192000))).astype(np.float32))
    mel = htsat_ref.logmel(wave)
    m = hf_audio_model(sd)
    taps = []
    with torch.no_grad():
        hf = m(input_features=mel.unsqueeze(1), output_hidden_states=True)
        latent = htsat_ref.body_forward(sd, mel, taps=taps)
        out = htsat_ref.htsat_forward(sd, wave)
    d_lat = (hf.pooler_output - latent).abs().max().item()
    big = max(t.abs().max().item() for t in taps)
    print(f"  pin body (checkpoint-like): |oracle - HF| pooled latent {d_lat:.3e} (scale {latent.abs().max():.2f}); "
          f"largest residual value {big:.1f}")
    assert d_lat < 5e-4 * max(1.0, latent.abs().max().item())
    np.savez_compressed(GOLD / "htsat_stress.npz", out=out.numpy(), latent=latent.numpy(), weight_seed=5, wave_seed=14,
                        pin_latent=d_lat, largest_residual=big, tap4=taps[-1][:, :4, :].numpy())
    print(f"  wrote htsat_stress.npz: out {tuple(out.shape)}")


def main():
    GOLD.mkdir(parents=True, exist_ok=True)
    torch.set_num_threads(8)
    if "--stress-only" in sys.argv:
        stress()
        return
    stress()
    sd = random_htsat_state_dict(0)
    rng = np.random.default_rng(4)
    wave = torch.from_numpy((0.1 * rng.standard_normal((2, 192000))).astype(np.float32))  # 4 s @ 48 kHz (reference)
    # 1. STFT pin: explicit DFT vs torch.stft
    p_ref = torch.stft(wave, n_fft=1024, hop_length=320, win_length=1024, window=torch.hann_window(1024, periodic=True),
                       center=True, pad_mode="reflect", return_complex=True).abs().pow(2).transpose(1, 2)
    p_mine = htsat_ref.power_spectrogram(wave)
    d_stft = ((p_mine - p_ref).abs().max() / p_ref.abs().max()).item()
    print(f"  pin STFT power: frames {p_mine.shape[1]}, rel diff vs torch.stft {d_stft:.2e}")
    assert p_mine.shape == p_ref.shape == (2, 601, 513) and d_stft < 1e-4
    mel = htsat_ref.logmel(wave)
    # 2. body pin vs HF ClapAudioModel (expects [B,1,T,64] log-mel; BN etc. inside)
    m = hf_audio_model(sd)
    taps = []
    with torch.no_grad():
        hf = m(input_features=mel.unsqueeze(1), output_hidden_states=True)
        latent = htsat_ref.body_forward(sd, mel, taps=taps)
    d_lat = (hf.pooler_output - latent).abs().max().item()
    print(f"  pin body: |oracle - HF| pooled latent {d_lat:.3e} (scale {latent.abs().max():.2f})")
    assert d_lat < 5e-4 * max(1.0, latent.abs().max().item())
    with torch.no_grad():
        out = htsat_ref.htsat_forward(sd, wave)
        # 10-s clip path (1501 frames -> first 1024 frames used)
        wave10 = torch.from_numpy((0.1 * rng.standard_normal((1, 480000))).astype(np.float32))
        out10 = htsat_ref.htsat_forward(sd, wave10)
    tap_rows = [t[:, :4, :].numpy() for t in taps]  # first 4 tokens of every tap (patch embed + 4 stages)
    np.savez_compressed(GOLD / "htsat.npz", out=out.numpy(), out10=out10.numpy(), latent=latent.numpy(),
                        mel_head=mel[:, :8, :].numpy(), weight_seed=0, wave_seed=4, pin_latent=d_lat, pin_stft=d_stft,
                        **{f"tap{i}": t for i, t in enumerate(tap_rows)})
    print(f"  wrote htsat.npz: out {tuple(out.shape)} norm {out.norm(dim=1)}")


if __name__ == "__main__":
    main()
