"""SigLIP towers (the model of the reference's end-to-end test, tests/test-kinetics-6.sh:91) without a GPU: the oracles
against their committed golden vectors (pinned to transformers' Siglip when the fixtures were made), weight packing against
the layout the library reports, the squash transform, and the tokenizer restatement."""
import ctypes as C
from pathlib import Path

import numpy as np
import pytest
import torch
from PIL import Image

from oracle import siglip_ref
from oracle.make_golden_siglip import TINY_T, TINY_V, TINY_V_TANH, normalize_u8, seeded_frames, seeded_tokens
from wise_amd import _lib
from wise_amd.feature.siglip import (SIGLIP_TEXT, SIGLIP_VISION, SiglipTokenizer, canonicalize, pack_siglip_text,
                                     pack_siglip_vision, random_siglip_text_state_dict, random_siglip_vision_state_dict,
                                     siglip_text_keys, siglip_vision_keys)

GOLD = Path(__file__).parent / "golden"


@pytest.mark.parametrize("spec,fname", [(TINY_V, "siglip_v_tiny.npz"), (TINY_V_TANH, "siglip_v_tiny100.npz")])
def test_vision_oracle_reproduces_golden(spec, fname):
    g = np.load(GOLD / fname)
    seed, n, fseed = (int(v) for v in g["meta"])
    sd = random_siglip_vision_state_dict(spec, seed)
    x = normalize_u8(torch.from_numpy(seeded_frames(n, spec.image_size, fseed)))
    taps = []
    with torch.no_grad():
        out = siglip_ref.siglip_vision_forward(sd, x, patch=spec.patch, heads=spec.heads, act=spec.act, taps=taps)
    assert np.allclose(out.numpy(), g["out"], atol=2e-6) and np.allclose(out.norm(dim=1).numpy(), 1.0, atol=1e-6)
    assert np.allclose(np.stack([t[:, 0, :].numpy() for t in taps]), g["taps"], atol=5e-5)
    assert float(g["pin_hidden"]) < 1e-3 and (spec.act != "gelu" or float(g["pin_out"]) < 1e-4)


def test_text_oracle_reproduces_golden_and_is_not_causal():
    spec = TINY_T
    g = np.load(GOLD / "siglip_t_tiny.npz")
    seed, n, tseed = (int(v) for v in g["meta"])
    tokens = torch.from_numpy(seeded_tokens(n, spec, tseed))
    assert np.array_equal(tokens.numpy(), g["tokens"])
    sd = random_siglip_text_state_dict(spec, seed)
    with torch.no_grad():
        out = siglip_ref.siglip_text_forward(sd, tokens, heads=spec.heads, act=spec.act)
        t2 = tokens.clone(); t2[0, 0] = 7        # the FIRST token reaches the pooled LAST position: no causal mask in the way
        out2 = siglip_ref.siglip_text_forward(sd, t2, heads=spec.heads, act=spec.act)
    assert np.allclose(out.numpy(), g["out"], atol=2e-6)
    assert float(g["pin_out"]) < 1e-4 and float(g["pin_hidden"]) < 1e-3
    assert (out[0] - out2[0]).abs().max() > 1e-3 and torch.allclose(out[1:], out2[1:])


def test_weight_layouts_match_library():
    lib = _lib.load()
    for spec in (TINY_V, TINY_V_TANH, SIGLIP_VISION["ViT-L-16-SigLIP-384"], SIGLIP_VISION["ViT-B-16-SigLIP-256"]):
        cfg = spec.c_config()
        nb, nf = C.c_int64(), C.c_int64()
        assert lib.wise_vit_layout(C.byref(cfg), C.byref(nb), C.byref(nf)) == 0
        W, F, L, T = spec.width, spec.mlp, spec.layers, spec.tokens
        assert T == (spec.image_size // spec.patch) ** 2                          # no class token
        assert nb.value == W * spec.kpad + (L + 1) * (4 * W * W + 2 * F * W) - W * W     # head: kv is 2W x W, no q matrix
        assert nf.value == T * W + L * (9 * W + F) + 2 * W + (7 * W + F)
        assert lib.wise_vit_workspace_bytes(C.byref(cfg), 3) > 0
    sd = random_siglip_vision_state_dict(TINY_V, 0)
    assert list(sd) == [k for k, _ in siglip_vision_keys(TINY_V)]
    wb, pf = pack_siglip_vision(TINY_V, sd)
    cfg = TINY_V.c_config()
    nb, nf = C.c_int64(), C.c_int64()
    lib.wise_vit_layout(C.byref(cfg), C.byref(nb), C.byref(nf))
    assert wb.numel() == nb.value and pf.numel() == nf.value
    for spec in (TINY_T, SIGLIP_TEXT["ViT-L-16-SigLIP-384"]):
        cfg = spec.c_config()
        assert (cfg.pool, cfg.head, cfg.no_causal, cfg.eps_e6) == (2, 2, 1, 1)
        assert lib.wise_text_layout(C.byref(cfg), C.byref(nb), C.byref(nf)) == 0
        W, F, L, D = spec.width, spec.mlp, spec.layers, spec.embed_dim
        assert nb.value == L * (4 * W * W + 2 * F * W) + D * W
        assert nf.value == (spec.vocab + spec.context) * W + L * (9 * W + F) + 2 * W + D
    sd = random_siglip_text_state_dict(TINY_T, 0)
    assert list(sd) == [k for k, _ in siglip_text_keys(TINY_T)]
    wb, pf = pack_siglip_text(TINY_T, sd)
    cfg = TINY_T.c_config()
    lib.wise_text_layout(C.byref(cfg), C.byref(nb), C.byref(nf))
    assert wb.numel() == nb.value and pf.numel() == nf.value
    bad = _lib.VitConfig(64, 16, 128, 2, 2, 256, 64, 1, 1)             # the attention-pool head has no projection: D == W
    assert lib.wise_vit_layout(C.byref(bad), C.byref(nb), C.byref(nf)) != 0


def test_squash_transform_and_feature_ids():
    from wise_amd.feature.mlfoundation_openclip import ClipImageTransform, list_pretrained

    assert ("ViT-L-16-SigLIP-384", "webli") in list_pretrained() and ("ViT-B-16-SigLIP-256", "webli") in list_pretrained()
    img = Image.fromarray(np.random.default_rng(0).integers(0, 256, size=(120, 200, 3), dtype=np.uint8))
    x = ClipImageTransform(64, (0.5,) * 3, (0.5,) * 3, "squash")(img)
    ref = torch.from_numpy(np.asarray(img.resize((64, 64), Image.BICUBIC), dtype=np.uint8).copy()).permute(2, 0, 1).float()
    assert x.shape == (3, 64, 64) and torch.allclose(x, (ref / 255.0 - 0.5) / 0.5)
    assert float(x.min()) >= -1.0 and float(x.max()) <= 1.0


def test_tokenizer_restates_canonicalize_plus_sentencepiece(tmp_path):
    import sentencepiece as spm

    rng = np.random.default_rng(0)
    words = ["".join(rng.choice(list("abcdefghijklmnopqrstuvwxyz"), int(rng.integers(2, 9)))) for _ in range(300)]
    (tmp_path / "c.txt").write_text("\n".join(" ".join(rng.choice(words, 8)) for _ in range(2000)))
    # T5-style vocabulary: <pad> = 0, </s> = 1, <unk> = 2, no <s>
    spm.SentencePieceTrainer.train(input=str(tmp_path / "c.txt"), model_type="unigram", vocab_size=300, minloglevel=2,
                                   model_prefix=str(tmp_path / "spiece"), pad_id=0, eos_id=1, unk_id=2, bos_id=-1)
    tk = SiglipTokenizer(tmp_path / "spiece.model", context=64)
    assert canonicalize("  A photo, of: Dogs & cats!! ") == "a photo of dogs cats"
    assert canonicalize("snake_case and   spaces") == "snake case and spaces"
    sp = spm.SentencePieceProcessor(model_file=str(tmp_path / "spiece.model"))
    texts = [" ".join(words[:4]).upper() + "!", "", " ".join(rng.choice(words, 200))]
    tok = tk(texts)
    assert tok.shape == (3, 64) and tok.dtype == torch.int64
    ids0 = sp.encode(canonicalize(texts[0]))
    assert tok[0, : len(ids0)].tolist() == ids0 and int(tok[0, len(ids0)]) == 1 and (tok[0, len(ids0):] == 1).all()
    assert int(tok[1, 0]) == 1 and (tok[1] == 1).all()                 # empty text: </s> then padding (same id)
    assert int(tok[2, 63]) == 1 and (tok[2, :63] != 1).all()           # truncated: 63 pieces + </s>
    # transformers' T5 tokenizer on the same model and the same cleaned text yields the same ids before padding
    try:
        from transformers import T5Tokenizer
        hf = T5Tokenizer.from_pretrained(str(tmp_path), legacy=False)
        hf_ids = hf(canonicalize(texts[0])).input_ids
        assert hf_ids == ids0 + [1]
    except Exception as e:  # the slow tokenizer's loader differs between transformers releases
        pytest.skip(f"transformers T5Tokenizer not loadable here: {type(e).__name__}")
