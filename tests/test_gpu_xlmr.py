"""-m gpu: XLM-RoBERTa text tower parity (the text side of the reference's default model pair, extract-features.py:192) —
wise_xlmr_forward through the C ABI against the fp32 CPU oracle (pinned to transformers' XLMRobertaModel) and the committed
golden vectors.  Tolerance (BASELINE.json north_star): cosine within 1e-3 of the fp32 path."""
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import vit_ref, xlmr_text_ref
from oracle.make_golden_xlmr import TINY, TINY_SHORT, seeded_tokens
from wise_amd import _lib
from wise_amd.feature.xlmr_text import XLMR_SPECS, XlmrTextEngine, random_xlmr_state_dict

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).parent / "golden"
COS_TOL = 1e-3


def cosine(a, b):
    a = a.double(); b = b.double()
    return ((a * b).sum(-1) / (a.norm(dim=-1) * b.norm(dim=-1))).min().item()


@pytest.mark.parametrize("B,T,H", [(1, 1, 2), (3, 5, 2), (2, 64, 8), (4, 77, 16), (2, 128, 4), (5, 65, 2), (3, 200, 4)])
def test_attention_with_padded_keys(B, T, H):
    """keys past lens[b] are invisible to every query of sequence b; the rows of real tokens equal attention over the
    sequence cut to its own length"""
    lib = _lib.lib()
    g = torch.Generator().manual_seed(B * 1000 + T * 10 + H)
    qkv = (torch.randn(B * T, 3 * H * 64, generator=g) * 1.5).to(torch.bfloat16)
    lens = torch.randint(1, T + 1, (B,), generator=g).to(torch.int32)
    lens[0] = T
    qd, ld = qkv.cuda(), lens.cuda()
    out = torch.full((B * T, H * 64), float("nan"), dtype=torch.bfloat16, device="cuda")
    from ctypes import c_void_p
    rc = lib.wise_attention_lens_bf16(qd.data_ptr(), B, T, H, ld.data_ptr(), out.data_ptr(), _lib.stream_ptr())
    _lib.check(rc, "attn")
    torch.cuda.synchronize()
    got = out.float().cpu().reshape(B, T, H * 64)
    assert torch.isfinite(got).all()
    for b in range(B):
        n = int(lens[b])
        ref = vit_ref.attention_ref(qkv.float().reshape(B, T, -1)[b, :n], 1, n, H)
        assert (got[b, :n] - ref).abs().max().item() < 3e-2


@pytest.mark.parametrize("spec,fname", [(TINY, "xlmr_tiny.npz"), (TINY_SHORT, "xlmr_tiny_short.npz"),
                                        (XLMR_SPECS["xlm-roberta-large-ViT-H-14"], "xlmr_large.npz"),
                                        (XLMR_SPECS["xlm-roberta-base-ViT-B-32"], "xlmr_base.npz")])
def test_xlmr_forward_matches_golden(spec, fname):
    gold = np.load(GOLD / fname)
    seed, n, tok_seed = (int(v) for v in gold["meta"][:3])
    tokens = seeded_tokens(n, spec, tok_seed)
    assert np.array_equal(tokens, gold["tokens"])
    eng = XlmrTextEngine(spec, random_xlmr_state_dict(spec, seed), max_batch=n)
    out = eng.forward(torch.from_numpy(tokens))
    torch.cuda.synchronize()
    got, want = out.cpu(), torch.from_numpy(gold["out"])
    assert got.shape == want.shape
    assert abs(got.norm(dim=1) - 1).max() < 1e-5
    assert cosine(got, want) > 1 - COS_TOL, cosine(got, want)
    # hidden state of the <s> row of every sequence after the last layer
    res = eng.residual(n).cpu().reshape(n, spec.context, spec.width)[:, 0]
    assert cosine(res, torch.from_numpy(gold["taps"][-1])) > 1 - 2e-3


def test_padding_batch_independence_and_argument_checks():
    spec = TINY
    sd = random_xlmr_state_dict(spec, 3)
    eng = XlmrTextEngine(spec, sd, max_batch=8)
    tok = np.full((4, spec.context), spec.pad_id, dtype=np.int32)
    tok[0, :4] = [0, 5, 6, 2]
    tok[1, :2] = [0, 2]                                # empty text
    tok[2, :] = 7; tok[2, 0] = 0; tok[2, -1] = 2       # full context
    tok[3, :9] = [0, 9, 8, 7, 6, 5, 4, 10, 2]
    t = torch.from_numpy(tok)
    with torch.no_grad():
        want = xlmr_text_ref.xlmr_text_forward(sd, t, heads=spec.heads, pad_id=spec.pad_id)
    got = eng.forward(t).cpu()
    assert cosine(got, want) > 1 - COS_TOL
    # a row's embedding does not depend on its neighbours in the batch (a single query takes the split-K kernels, whose
    # fixed summation order differs from the batched tiles': equal to rounding, and bit-stable from call to call)
    alone = torch.cat([eng.forward(t[i:i + 1]).cpu() for i in range(4)])
    assert cosine(alone, got) > 1 - 1e-5 and (alone - got).abs().max() < 1e-3      # bf16 roundings flip here and there
    assert torch.equal(torch.cat([eng.forward(t[i:i + 1]).cpu() for i in range(4)]), alone)
    big = eng.forward(t.repeat(70, 1)).cpu()           # 280 rows: more than one 256-row GEMM tile of pooled rows
    assert cosine(big[:4], got) > 1 - 1e-5 and torch.equal(big[:4], big[-4:])
    # bidirectional: a later token matters to the whole sequence
    tok2 = tok.copy(); tok2[3, 7] = 11
    assert (eng.forward(torch.from_numpy(tok2)).cpu()[3] - got[3]).abs().max() > 1e-3
    bad = tok.copy(); bad[0, 2] = spec.pad_id          # padding in the middle: not what the tokenizer produces
    with pytest.raises(ValueError, match="right-padded"):
        eng.forward(torch.from_numpy(bad))
    with pytest.raises(ValueError, match="vocabulary"):
        eng.forward(torch.from_numpy(np.where(tok == 9, spec.vocab, tok)))


def test_default_model_pair_end_to_end(tmp_path, monkeypatch):
    """FeatureExtractorFactory on the reference's default id with seeded weights: image and text land in one 1024-d space;
    the tokenizer is the model's sentencepiece vocabulary (a small one trained here)"""
    import sentencepiece as spm

    from wise_amd.feature.feature_extractor_factory import FeatureExtractorFactory

    rng = np.random.default_rng(0)
    words = ["".join(rng.choice(list("abcdefghijklmnopqrstuvwxyz"), int(rng.integers(2, 9)))) for _ in range(300)]
    (tmp_path / "c.txt").write_text("\n".join(" ".join(rng.choice(words, 8)) for _ in range(2000)))
    (tmp_path / "xlm-roberta-large").mkdir()
    spm.SentencePieceTrainer.train(input=str(tmp_path / "c.txt"), model_type="unigram", vocab_size=300, minloglevel=2,
                                   model_prefix=str(tmp_path / "xlm-roberta-large" / "sentencepiece.bpe"))
    monkeypatch.setenv("WISE_AMD_WEIGHTS_DIR", str(tmp_path))
    fx = FeatureExtractorFactory("mlfoundations/open_clip/xlm-roberta-large-ViT-H-14/seeded-0")
    assert fx.get_output_dim() == 1024 and fx.get_input_image_size() == (224, 224)
    texts = [" ".join(words[:3]), " ".join(words[3:12])]
    feats = fx.extract_text_features(texts)
    assert feats.shape == (2, 1024) and np.allclose(np.linalg.norm(feats, axis=1), 1.0, atol=1e-5)
    tok = fx.preprocess_text(texts)
    assert tok.shape == (2, 77) and int(tok[0, 0]) == 0 and int(tok[0, 4]) in (2, tok[0, 4]) and (tok[:, -1] == 1).all()
    from wise_amd.feature.xlmr_text import random_xlmr_state_dict as rsd
    with torch.no_grad():
        want = xlmr_text_ref.xlmr_text_forward(rsd(fx.text_spec, 0), tok, heads=fx.text_spec.heads)
    assert cosine(torch.from_numpy(feats), want) > 1 - COS_TOL
