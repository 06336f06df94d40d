"""ORACLE tooling: pin oracle/vit_ref.py against transformers' CLIP (an independent implementation of
the same published architecture, present in the authoring container) and write the committed
fixtures under tests/golden/.  Run in the authoring container only:  python -m oracle.make_golden

Fixtures are DATA (inputs are regenerated from seeds; expected outputs are stored):
  tests/golden/vit_tiny.npz     tiny ViT (W=128, L=2): full taps + output, fp32
  tests/golden/vit_b32.npz      ViT-B/32, seed-0 weights, 4 seeded frames: cls-token row of every tap + [4,512]
  tests/golden/vit_l14.npz      ViT-L/14, seed-0 weights, 2 seeded frames: cls rows + [2,768]
  tests/golden/ip_topk.npz      X [4096,512], Q [8,512], k in {1,10,100}: D, I ; tie case ; N<k case
"""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from oracle import ip_topk_ref, vit_ref  # noqa: E402
from wise_amd.feature.vit import VitSpec, checkpoint_like_state_dict, random_state_dict, spec_for  # noqa: E402

GOLD = ROOT / "tests" / "golden"

TINY = VitSpec("tiny", 64, 32, 128, 2, 2, 512, 64, "quick_gelu")
TINY_GELU = VitSpec("tiny-gelu", 56, 14, 128, 1, 2, 256, 32, "gelu")
# head width 80 as in ViT-H/14 (width 640 over 8 heads), 9 x 9 patches + class token = 82 tokens: two key blocks
TINY_H80 = VitSpec("tiny-h80", 126, 14, 640, 2, 8, 1280, 64, "gelu")


def seeded_frames(n: int, S: int, seed: int) -> np.ndarray:
    return np.random.default_rng(seed).integers(0, 256, size=(n, 3, S, S), dtype=np.uint8)


def hf_model(spec: VitSpec, sd):
    from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection

    cfg = CLIPVisionConfig(hidden_size=spec.width, intermediate_size=spec.mlp, num_hidden_layers=spec.layers,
                           num_attention_heads=spec.heads, image_size=spec.image_size, patch_size=spec.patch,
                           projection_dim=spec.embed_dim, hidden_act=spec.act, attention_dropout=0.0,
                           layer_norm_eps=1e-5)
    m = CLIPVisionModelWithProjection(cfg).eval()
    W = spec.width
    new = {}
    new["vision_model.embeddings.class_embedding"] = sd["visual.class_embedding"]
    new["vision_model.embeddings.patch_embedding.weight"] = sd["visual.conv1.weight"]
    new["vision_model.embeddings.position_embedding.weight"] = sd["visual.positional_embedding"]
    new["vision_model.pre_layrnorm.weight"] = sd["visual.ln_pre.weight"]
    new["vision_model.pre_layrnorm.bias"] = sd["visual.ln_pre.bias"]
    for i in range(spec.layers):
        p = f"visual.transformer.resblocks.{i}."
        h = f"vision_model.encoder.layers.{i}."
        wq, wk, wv = sd[p + "attn.in_proj_weight"].split(W, dim=0)
        bq, bk, bv = sd[p + "attn.in_proj_bias"].split(W, dim=0)
        for n, w_, b_ in (("q", wq, bq), ("k", wk, bk), ("v", wv, bv)):
            new[h + f"self_attn.{n}_proj.weight"] = w_
            new[h + f"self_attn.{n}_proj.bias"] = b_
        new[h + "self_attn.out_proj.weight"] = sd[p + "attn.out_proj.weight"]
        new[h + "self_attn.out_proj.bias"] = sd[p + "attn.out_proj.bias"]
        new[h + "layer_norm1.weight"] = sd[p + "ln_1.weight"]
        new[h + "layer_norm1.bias"] = sd[p + "ln_1.bias"]
        new[h + "layer_norm2.weight"] = sd[p + "ln_2.weight"]
        new[h + "layer_norm2.bias"] = sd[p + "ln_2.bias"]
        new[h + "mlp.fc1.weight"] = sd[p + "mlp.c_fc.weight"]
        new[h + "mlp.fc1.bias"] = sd[p + "mlp.c_fc.bias"]
        new[h + "mlp.fc2.weight"] = sd[p + "mlp.c_proj.weight"]
        new[h + "mlp.fc2.bias"] = sd[p + "mlp.c_proj.bias"]
    new["vision_model.post_layernorm.weight"] = sd["visual.ln_post.weight"]
    new["vision_model.post_layernorm.bias"] = sd["visual.ln_post.bias"]
    new["visual_projection.weight"] = sd["visual.proj"].t().contiguous()
    missing, unexpected = m.load_state_dict(new, strict=False)
    missing = [k for k in missing if "position_ids" not in k]
    assert not missing and not unexpected, (missing, unexpected)
    return m


def pin_against_hf(spec: VitSpec, sd, x: torch.Tensor, tol: float):
    """oracle vs transformers CLIP on the same weights/inputs: un-normalised embeddings + hidden states."""
    m = hf_model(spec, sd)
    with torch.no_grad():
        hf = m(pixel_values=x, output_hidden_states=True)
        taps = []
        ours = vit_ref.vit_forward(sd, x, patch=spec.patch, heads=spec.heads, act=spec.act, taps=taps,
                                   normalize=False)
    d_out = (hf.image_embeds - ours).abs().max().item()
    # HF hidden_states[0] is the embedding BEFORE pre_layrnorm; [i>=1] are after block i
    d_hid = max((hf.hidden_states[i] - taps[i]).abs().max().item() for i in range(1, spec.layers + 1)) \
        if spec.layers else 0.0
    scale = ours.abs().max().item()
    print(f"  pin {spec.name}: |oracle - HF| out {d_out:.3e} (scale {scale:.2f}), hidden {d_hid:.3e}")
    hid_scale = max(t.abs().max().item() for t in taps) if taps else 1.0
    assert d_out <= tol * max(scale, 1.0) and d_hid <= tol * max(50.0, hid_scale), "oracle does not match transformers CLIP"
    return d_out, d_hid


def golden_vit(spec: VitSpec, seed: int, n_frames: int, frame_seed: int, fname: str, full_taps: bool, pin: bool,
               checkpoint_like: bool = False):
    print(f"[vit] {spec.name}" + (" (checkpoint-like weights)" if checkpoint_like else ""))
    sd = checkpoint_like_state_dict(spec, seed) if checkpoint_like else random_state_dict(spec, seed)
    frames = seeded_frames(n_frames, spec.image_size, frame_seed)
    x = vit_ref.normalize_u8(torch.from_numpy(frames))
    torch.set_num_threads(8)
    pinned = (float("nan"), float("nan"))
    if pin:
        pinned = pin_against_hf(spec, sd, x, 2e-5)
    taps = []
    with torch.no_grad():
        out = vit_ref.vit_forward(sd, x, patch=spec.patch, heads=spec.heads, act=spec.act, taps=taps)
    taps_np = np.stack([t.numpy() if full_taps else t[:, 0, :].numpy() for t in taps])
    np.savez_compressed(GOLD / fname, out=out.numpy(), taps=taps_np, weight_seed=seed, frame_seed=frame_seed,
                        n_frames=n_frames, pin_out=pinned[0], pin_hidden=pinned[1],
                        spec=np.array([spec.image_size, spec.patch, spec.width, spec.layers, spec.heads, spec.mlp,
                                       spec.embed_dim, 0 if spec.act == "quick_gelu" else 1]))
    print(f"  wrote {fname}: out {out.shape}, taps {taps_np.shape}")


def golden_ip():
    print("[ip_topk]")
    rng = np.random.default_rng(2)
    X = rng.standard_normal((4096, 512), dtype=np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    Q = np.random.default_rng(3).standard_normal((8, 512), dtype=np.float32)
    Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    ids = np.arange(4096, dtype=np.int64) + 1  # sqlite autoincrement starts at 1
    out = {}
    for k in (1, 10, 100):
        D, I = ip_topk_ref.ip_topk(X, Q, k, ids=ids)
        out[f"D{k}"], out[f"I{k}"] = D, I
    # tie case: rows 5, 17, 900 identical (black frames give identical embeddings) and best for query 0
    Xt = X[:1024].copy()
    Xt[[5, 17, 900]] = Q[0]
    Dt, It = ip_topk_ref.ip_topk(Xt, Q[:2], 5, ids=ids[:1024])
    out["Dtie"], out["Itie"] = Dt, It
    # N < k
    Ds, Is = ip_topk_ref.ip_topk(X[:7], Q[:2], 10, ids=ids[:7])
    out["Dshort"], out["Ishort"] = Ds, Is
    np.savez_compressed(GOLD / "ip_topk.npz", **out)
    print("  wrote ip_topk.npz")


def main():
    GOLD.mkdir(parents=True, exist_ok=True)
    if "--stress-only" in sys.argv:
        golden_vit(TINY, 21, 3, 31, "vit_tiny_stress.npz", full_taps=True, pin=True, checkpoint_like=True)
        golden_vit(spec_for("ViT-B-32"), 3, 4, 32, "vit_b32_stress.npz", full_taps=False, pin=True, checkpoint_like=True)
        golden_vit(spec_for("ViT-L-14"), 4, 2, 33, "vit_l14_stress.npz", full_taps=False, pin=True, checkpoint_like=True)
        return
    golden_vit(TINY, 7, 3, 11, "vit_tiny.npz", full_taps=True, pin=True)
    golden_vit(TINY_GELU, 8, 2, 12, "vit_tiny_gelu.npz", full_taps=True, pin=True)
    golden_vit(TINY_H80, 9, 2, 13, "vit_tiny_h80.npz", full_taps=True, pin=True)
    golden_vit(spec_for("ViT-B-32"), 0, 4, 1, "vit_b32.npz", full_taps=False, pin=True)
    golden_vit(spec_for("ViT-L-14"), 0, 2, 5, "vit_l14.npz", full_taps=False, pin=True)
    golden_vit(spec_for("ViT-B-16", "laion2b_s34b_b88k"), 0, 2, 14, "vit_b16.npz", full_taps=False, pin=True)   # 197 tokens, erf GELU
    golden_vit(spec_for("ViT-H-14", "laion2b_s32b_b79k"), 0, 2, 6, "vit_h14.npz", full_taps=False, pin=True)
    # checkpoint-like statistics (massive activation channels, heavy-tailed LN gains, peaky attention): see
    # wise_amd/feature/vit.py::checkpoint_like_state_dict
    golden_vit(TINY, 21, 3, 31, "vit_tiny_stress.npz", full_taps=True, pin=True, checkpoint_like=True)
    golden_vit(spec_for("ViT-B-32"), 3, 4, 32, "vit_b32_stress.npz", full_taps=False, pin=True, checkpoint_like=True)
    golden_vit(spec_for("ViT-L-14"), 4, 2, 33, "vit_l14_stress.npz", full_taps=False, pin=True, checkpoint_like=True)
    golden_ip()


if __name__ == "__main__":
    main()
