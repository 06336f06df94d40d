"""XLM-RoBERTa text tower (the text side of the reference's default model pair, extract-features.py:192) without a GPU:
the oracle against its committed golden vectors, padding invariance, the host-side weight layout against what the
library reports, and the tokenizer restatement against transformers' XLMRobertaTokenizer on the same sentencepiece model."""
import ctypes as C
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import xlmr_text_ref
from oracle.make_golden_xlmr import TINY, TINY_SHORT, seeded_tokens
from wise_amd import _lib
from wise_amd.feature.xlmr_text import (XLMR_SPECS, XlmrTokenizer, pack_xlmr_weights, random_xlmr_state_dict,
                                        xlmr_state_dict_keys)

GOLD = Path(__file__).parent / "golden"


@pytest.mark.parametrize("spec,fname", [(TINY, "xlmr_tiny.npz"), (TINY_SHORT, "xlmr_tiny_short.npz")])
def test_oracle_reproduces_golden(spec, fname):
    gold = np.load(GOLD / fname)
    seed, n, tok_seed = (int(v) for v in gold["meta"][:3])
    tokens = torch.from_numpy(seeded_tokens(n, spec, tok_seed))
    assert np.array_equal(tokens.numpy(), gold["tokens"])
    sd = random_xlmr_state_dict(spec, seed)
    taps = []
    with torch.no_grad():
        out = xlmr_text_ref.xlmr_text_forward(sd, tokens, heads=spec.heads, pad_id=spec.pad_id, taps=taps)
    assert np.allclose(out.numpy(), gold["out"], atol=2e-6)
    assert np.allclose(np.stack([t[:, 0, :].numpy() for t in taps]), gold["taps"], atol=5e-5)
    assert np.allclose(out.norm(dim=1).numpy(), 1.0, atol=1e-6)
    # the pin against transformers' XLMRobertaModel recorded when the fixture was made
    assert float(gold["pin_out"]) < 2e-5 and float(gold["pin_hidden"]) < 1e-3


def test_oracle_ignores_what_lies_behind_the_padding_and_is_bidirectional():
    spec = TINY
    sd = random_xlmr_state_dict(spec, 2)
    a = torch.full((1, spec.context), spec.pad_id, dtype=torch.int64)
    a[0, :5] = torch.tensor([0, 10, 11, 12, 2])
    with torch.no_grad():
        ea = xlmr_text_ref.xlmr_text_forward(sd, a, heads=spec.heads)
        # the same text in a shorter context: padding rows change nothing (mean over the sequence's own tokens)
        eb = xlmr_text_ref.xlmr_text_forward(sd, a[:, :8], heads=spec.heads)
        c = a.clone()
        c[0, 3] = 13                               # a LATER token changes the embedding as much as an earlier one would
        ec = xlmr_text_ref.xlmr_text_forward(sd, c, heads=spec.heads)
    assert torch.allclose(ea, eb, atol=1e-6)
    assert (ea - ec).abs().max() > 1e-3


def test_weight_layout_matches_library():
    lib = _lib.load()  # host-only entry points: no GPU needed
    for spec in (TINY, TINY_SHORT, XLMR_SPECS["xlm-roberta-large-ViT-H-14"], XLMR_SPECS["xlm-roberta-base-ViT-B-32"]):
        cfg = spec.c_config()
        nb, nf = C.c_int64(), C.c_int64()
        assert lib.wise_xlmr_layout(C.byref(cfg), C.byref(nb), C.byref(nf)) == 0
        W, F, L, D, Hd = spec.width, spec.mlp, spec.layers, spec.embed_dim, spec.proj_hidden
        assert nb.value == L * (4 * W * W + 2 * F * W) + Hd * W + D * Hd
        assert nf.value == (spec.vocab + spec.max_positions + 3) * W + L * (9 * W + F)
        assert lib.wise_xlmr_workspace_bytes(C.byref(cfg), 1) > 0
    big = XLMR_SPECS["xlm-roberta-large-ViT-H-14"]
    assert (big.width, big.layers, big.heads, big.mlp, big.proj_hidden, big.embed_dim, big.vocab, big.max_positions) == \
        (1024, 24, 16, 4096, 1024, 1024, 250002, 514)
    sd = random_xlmr_state_dict(TINY, 0)
    assert list(sd) == [k for k, _ in xlmr_state_dict_keys(TINY)]
    wb, pf = pack_xlmr_weights(TINY, sd)
    cfg = TINY.c_config()
    nb, nf = C.c_int64(), C.c_int64()
    lib.wise_xlmr_layout(C.byref(cfg), C.byref(nb), C.byref(nf))
    assert wb.numel() == nb.value and pf.numel() == nf.value and wb.dtype == torch.bfloat16
    bad = _lib.XlmrConfig(77, 1000, 60, 256, 2, 4, 512, 128, 128, 1)     # 60 position rows cannot hold 77 tokens
    assert lib.wise_xlmr_layout(C.byref(bad), C.byref(nb), C.byref(nf)) != 0
    bad = _lib.XlmrConfig(77, 1000, 80, 200, 2, 4, 512, 128, 128, 1)     # width not heads * 64
    assert lib.wise_xlmr_layout(C.byref(bad), C.byref(nb), C.byref(nf)) != 0


@pytest.fixture(scope="module")
def tiny_sentencepiece(tmp_path_factory):
    """a small sentencepiece model trained on synthetic text (the real sentencepiece.bpe.model is not available offline);
    unigram, like XLM-RoBERTa's own (the file name notwithstanding), with sentencepiece's default <unk>/<s>/</s> = 0/1/2"""
    import sentencepiece as spm

    d = tmp_path_factory.mktemp("spm")
    rng = np.random.default_rng(0)
    words = ["".join(rng.choice(list("abcdefghijklmnopqrstuvwxyzäöüéñ"), int(rng.integers(2, 9)))) for _ in range(400)]
    lines = [" ".join(rng.choice(words, int(rng.integers(3, 12)))) for _ in range(3000)]
    (d / "corpus.txt").write_text("\n".join(lines), encoding="utf-8")
    spm.SentencePieceTrainer.train(input=str(d / "corpus.txt"), model_prefix=str(d / "sentencepiece.bpe"), vocab_size=400,
                                   model_type="unigram", character_coverage=1.0, minloglevel=2)
    (d / "corpus.txt").unlink()
    return d / "sentencepiece.bpe.model", words


def test_tokenizer_matches_transformers_on_the_same_sentencepiece_model(tiny_sentencepiece):
    from transformers import XLMRobertaTokenizer

    model, words = tiny_sentencepiece
    hf = XLMRobertaTokenizer.from_pretrained(str(model.parent))      # converts the sentencepiece model it finds there
    tk = XlmrTokenizer(model, context=77)
    rng = np.random.default_rng(1)
    texts = ["a person riding a horse", "  two   spaces &amp; an entity ", "ÄÖÜ ñandú", "",
             " ".join(rng.choice(words, 120)),                   # longer than the context: truncated, specials kept
             "unknown ✓ glyph"] + [" ".join(rng.choice(words, int(rng.integers(1, 30)))) for _ in range(20)]
    mine = tk(texts)
    from wise_amd.feature.xlmr_text import _clean
    ref = hf([_clean(t) for t in texts], return_tensors="pt", max_length=77, padding="max_length", truncation=True).input_ids
    assert mine.shape == (len(texts), 77) and torch.equal(mine, ref)
    assert (mine[:, 0] == 0).all() and int(mine[3, 1]) == 2 and int(mine[3, 2]) == 1      # empty text: <s> </s> <pad>...
    assert int(mine[4, 76]) == 2                                                           # truncated row ends with </s>
    assert tk.vocab_size == len(hf)
    with pytest.raises(FileNotFoundError):
        import os
        old = os.environ.pop("WISE_AMD_WEIGHTS_DIR", None)
        try:
            XlmrTokenizer.default()
        finally:
            if old is not None:
                os.environ["WISE_AMD_WEIGHTS_DIR"] = old
