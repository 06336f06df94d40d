"""-m gpu: SigLIP towers (ViT-L-16-SigLIP-384/webli is the video feature id of the reference's end-to-end test,
tests/test-kinetics-6.sh:91) — the timm image tower with the attention-pool head through wise_vit_forward (arch 1) and the
non-causal text tower through wise_text_forward, against the fp32 CPU oracles (pinned to transformers' Siglip) and the
committed golden vectors.  Tolerance (BASELINE.json north_star): cosine within 1e-3 of the fp32 path."""
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import siglip_ref
from oracle.make_golden_siglip import TINY_T, TINY_V, TINY_V_TANH, normalize_u8, seeded_frames, seeded_tokens
from wise_amd.feature.siglip import (SIGLIP_TEXT, SIGLIP_VISION, pack_siglip_text, random_siglip_text_state_dict,
                                     random_siglip_vision_state_dict)
from wise_amd.feature.text import TextEngine
from wise_amd.feature.vit import VitEngine

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).parent / "golden"
COS_TOL = 1e-3


def cosine(a, b):
    a = a.double(); b = b.double()
    return ((a * b).sum(-1) / (a.norm(dim=-1) * b.norm(dim=-1))).min().item()


@pytest.mark.parametrize("spec,fname", [(TINY_V, "siglip_v_tiny.npz"), (TINY_V_TANH, "siglip_v_tiny100.npz"),
                                        (SIGLIP_VISION["ViT-B-16-SigLIP-256"], "siglip_v_b16_256.npz"),
                                        (SIGLIP_VISION["ViT-L-16-SigLIP-384"], "siglip_v_l16_384.npz")])
def test_vision_tower_matches_golden(spec, fname):
    g = np.load(GOLD / fname)
    seed, n, fseed = (int(v) for v in g["meta"])
    frames = torch.from_numpy(seeded_frames(n, spec.image_size, fseed))
    eng = VitEngine(spec, random_siglip_vision_state_dict(spec, seed), max_batch=max(n, 4))
    out_f32 = eng.forward(normalize_u8(frames).cuda()).cpu()
    out_u8 = eng.forward(frames.cuda()).cpu()                       # (x/255 - 0.5)/0.5 fused into the patch gather
    want = torch.from_numpy(g["out"])
    assert out_f32.shape == want.shape and abs(out_f32.norm(dim=1) - 1).max() < 1e-5
    assert cosine(out_f32, want) > 1 - COS_TOL, cosine(out_f32, want)
    assert cosine(out_u8, want) > 1 - COS_TOL
    # residual stream after the last block, first token of every frame
    res = eng.residual(n).cpu().reshape(n, spec.tokens, spec.width)[:, 0]
    assert cosine(res, torch.from_numpy(g["taps"][-1])) > 1 - 2e-3
    # a frame's embedding does not depend on its neighbours; one-stream and two-stream forms agree
    alone = torch.cat([eng.forward(frames[i:i + 1].cuda()).cpu() for i in range(n)])
    assert cosine(alone, out_u8) > 1 - 1e-6
    assert torch.equal(eng.forward(frames.cuda(), single_stream=True).cpu(), out_u8)


def test_vision_tower_batch_of_70_against_the_oracle():
    """more images than one 256-row tile of pooled rows would need padding for; every embedding against the oracle"""
    spec = TINY_V_TANH
    sd = random_siglip_vision_state_dict(spec, 9)
    frames = torch.from_numpy(seeded_frames(70, spec.image_size, 41))
    eng = VitEngine(spec, sd, max_batch=70)
    got = eng.forward(frames.cuda()).cpu()
    with torch.no_grad():
        want = siglip_ref.siglip_vision_forward(sd, normalize_u8(frames), patch=spec.patch, heads=spec.heads, act=spec.act)
    assert cosine(got, want) > 1 - COS_TOL
    pend = eng.forward_pipelined(frames.cuda())
    assert torch.equal(pend.result().cpu(), got)


@pytest.mark.parametrize("spec,fname", [(TINY_T, "siglip_t_tiny.npz"), (SIGLIP_TEXT["ViT-L-16-SigLIP-384"], "siglip_t_l16_384.npz")])
def test_text_tower_matches_golden(spec, fname):
    g = np.load(GOLD / fname)
    seed, n, tseed = (int(v) for v in g["meta"])
    tokens = seeded_tokens(n, spec, tseed)
    assert np.array_equal(tokens, g["tokens"])
    eng = TextEngine(spec, random_siglip_text_state_dict(spec, seed), max_batch=n, pack=pack_siglip_text)
    got = eng.forward(torch.from_numpy(tokens)).cpu()
    want = torch.from_numpy(g["out"])
    assert got.shape == want.shape and abs(got.norm(dim=1) - 1).max() < 1e-5
    assert cosine(got, want) > 1 - COS_TOL, cosine(got, want)
    res = eng.residual(n).cpu().reshape(n, spec.context, spec.width)[:, -1]
    assert cosine(res, torch.from_numpy(g["taps"][-1])) > 1 - 2e-3
    # no causal mask: changing the first token changes the embedding pooled at the last position
    t2 = tokens.copy(); t2[0, 0] = 5 if tokens[0, 0] != 5 else 6
    got2 = eng.forward(torch.from_numpy(t2)).cpu()
    assert (got2[0] - got[0]).abs().max() > 1e-3 and torch.equal(got2[1:], got[1:])


def test_reference_test_model_end_to_end(tmp_path, monkeypatch):
    """FeatureExtractorFactory on the video feature id of tests/test-kinetics-6.sh:91 with seeded weights: squash transform,
    384-pixel frames, 1024-d unit embeddings from both towers, text through the model's sentencepiece vocabulary"""
    import sentencepiece as spm
    from PIL import Image

    from wise_amd.feature.feature_extractor_factory import FeatureExtractorFactory
    from wise_amd.feature.siglip import random_siglip_text_state_dict as rtsd

    rng = np.random.default_rng(0)
    words = ["".join(rng.choice(list("abcdefghijklmnopqrstuvwxyz"), int(rng.integers(2, 9)))) for _ in range(300)]
    (tmp_path / "c.txt").write_text("\n".join(" ".join(rng.choice(words, 8)) for _ in range(2000)))
    (tmp_path / "siglip").mkdir()
    spm.SentencePieceTrainer.train(input=str(tmp_path / "c.txt"), model_type="unigram", vocab_size=300, minloglevel=2,
                                   model_prefix=str(tmp_path / "siglip" / "spiece"), pad_id=0, eos_id=1, unk_id=2, bos_id=-1)
    monkeypatch.setenv("WISE_AMD_WEIGHTS_DIR", str(tmp_path))
    fx = FeatureExtractorFactory("mlfoundations/open_clip/ViT-L-16-SigLIP-384/seeded-0")
    assert fx.get_output_dim() == 1024 and fx.get_input_image_size() == (384, 384)
    imgs = [Image.fromarray(rng.integers(0, 256, size=(240, 320, 3), dtype=np.uint8)) for _ in range(2)]
    x = fx.preprocess_image(imgs)
    assert tuple(x.shape) == (2, 3, 384, 384) and float(x.min()) >= -1 and float(x.max()) <= 1
    feats = fx.extract_image_features(x)
    assert feats.shape == (2, 1024) and np.allclose(np.linalg.norm(feats, axis=1), 1.0, atol=1e-5)
    texts = [" ".join(words[:3]), "A Photo, of " + words[5]]
    tfe = fx.extract_text_features(texts)
    assert tfe.shape == (2, 1024) and np.allclose(np.linalg.norm(tfe, axis=1), 1.0, atol=1e-5)
    tok = fx.preprocess_text(texts)
    with torch.no_grad():
        want = siglip_ref.siglip_text_forward(rtsd(fx.text_spec, 0), tok, heads=fx.text_spec.heads, act=fx.text_spec.act)
    assert cosine(torch.from_numpy(tfe), want) > 1 - COS_TOL
    # decoded uint8 frames through the GPU transform (squash) and the u8 input of the tower == the PIL path above
    raw = torch.from_numpy(np.stack([np.asarray(im) for im in imgs])).permute(0, 3, 1, 2).contiguous().cuda()
    u8 = fx.preprocess_image_device(raw)
    assert tuple(u8.shape) == (2, 3, 384, 384) and u8.dtype == torch.uint8
    assert torch.equal(((u8.cpu().float() / 255.0) - 0.5) / 0.5, x.cpu())
    feats_u8 = fx.extract_image_features(u8)
    assert cosine(torch.from_numpy(feats_u8), torch.from_numpy(feats)) > 1 - 1e-5
