"""MS-CLAP 2022 without a GPU: the caption encoder's oracle (pinned to transformers' BertModel) against its committed
golden vectors, the WordPiece tokenizer against transformers' BertTokenizer on the same vocabulary, the host-side weight
layouts (BERT caption encoder, Cnn14 audio encoder) against what the library reports, and the Cnn14 oracle against its
golden vectors and against BatchNorm folding as the packer does it."""
import ctypes as C
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import clap_bert_ref, cnn14_ref
from oracle.make_golden_clap_bert import TINY, seeded_tokens
from oracle.make_golden_cnn14 import golden_clips
from wise_amd import _lib
from wise_amd.feature.bert_tokenizer import BertTokenizer, basic_tokens, synthetic_vocab
from wise_amd.feature.clap_bert import (CLAP_BERT_SPEC, clap_bert_state_dict_keys, pack_clap_bert_weights,
                                        random_clap_bert_state_dict)
from wise_amd.feature.cnn14 import CHANNELS, pack_cnn14_weights, random_cnn14_state_dict, state_dict_keys

GOLD = Path(__file__).parent / "golden"


def test_bert_oracle_reproduces_golden():
    gold = np.load(GOLD / "clap_bert_tiny.npz")
    seed, n, tok_seed = (int(v) for v in gold["meta"][:3])
    tokens = torch.from_numpy(seeded_tokens(n, TINY, tok_seed))
    assert np.array_equal(tokens.numpy(), gold["tokens"])
    sd = random_clap_bert_state_dict(TINY, seed)
    taps = []
    with torch.no_grad():
        clap_bert_ref.bert_hidden(sd, tokens, heads=TINY.heads, taps=taps)
    out = clap_bert_ref.caption_forward_2022(sd, tokens, heads=TINY.heads)
    assert np.allclose(out.numpy(), gold["out"], atol=2e-6)
    assert np.allclose(np.stack([t[:, 0, :].numpy() for t in taps]), gold["taps"], atol=5e-5)
    assert np.allclose(out.norm(dim=1).numpy(), 1.0, atol=1e-6)
    # the pin against transformers' BertModel recorded when the fixtures were made (tiny and bert-base sizes)
    assert float(gold["pin_cls"]) < 5e-5 and float(np.load(GOLD / "clap_bert_base.npz")["pin_cls"]) < 5e-5
    # padding rows change nothing for the [CLS] row: the same text in a shorter context
    with torch.no_grad():
        short = clap_bert_ref.caption_forward_2022(sd, tokens[:1, :40], heads=TINY.heads)
    assert torch.allclose(short, out[:1], atol=1e-6)


def test_bert_weight_layout_matches_library():
    lib = _lib.load()  # host-only entry points: no GPU needed
    for spec in (TINY, CLAP_BERT_SPEC):
        cfg = spec.c_config()
        nb, nf = C.c_int64(), C.c_int64()
        assert lib.wise_xlmr_layout(C.byref(cfg), C.byref(nb), C.byref(nf)) == 0
        W, F, L, D = spec.width, spec.mlp, spec.layers, spec.embed_dim
        assert nb.value == L * (4 * W * W + 2 * F * W) + D * W + D * D
        assert nf.value == (spec.vocab + spec.max_positions + 3) * W + L * (9 * W + F) + 2 * D
        assert lib.wise_xlmr_workspace_bytes(C.byref(cfg), 1) > 0
    s = CLAP_BERT_SPEC
    assert (s.width, s.layers, s.heads, s.mlp, s.embed_dim, s.vocab, s.max_positions, s.context, s.pad_id) == \
        (768, 12, 12, 3072, 1024, 30522, 512, 100, 0)
    sd = random_clap_bert_state_dict(TINY, 0)
    assert list(sd) == [k for k, _ in clap_bert_state_dict_keys(TINY)]
    wb, pf = pack_clap_bert_weights(TINY, sd)
    cfg = TINY.c_config()
    nb, nf = C.c_int64(), C.c_int64()
    lib.wise_xlmr_layout(C.byref(cfg), C.byref(nb), C.byref(nf))
    assert wb.numel() == nb.value and pf.numel() == nf.value and wb.dtype == torch.bfloat16
    bad = _lib.XlmrConfig(100, 1536, 128, 256, 2, 4, 512, 512, 512, 0, 1, 1, 1, 1)   # msclap head needs 1024 -> 1024
    assert lib.wise_xlmr_layout(C.byref(bad), C.byref(nb), C.byref(nf)) != 0
    bad = _lib.XlmrConfig(100, 1536, 64, 256, 2, 4, 512, 1024, 1024, 0, 1, 1, 1, 1)  # 64 position rows, 100 tokens
    assert lib.wise_xlmr_layout(C.byref(bad), C.byref(nb), C.byref(nf)) != 0
    bad = _lib.XlmrConfig(100, 1536, 128, 256, 2, 4, 512, 1024, 1024, 0, 2, 1, 1, 1)  # unknown pos_mode
    assert lib.wise_xlmr_layout(C.byref(bad), C.byref(nb), C.byref(nf)) != 0


@pytest.fixture(scope="module")
def word_vocab(tmp_path_factory):
    """a WordPiece vocabulary with whole words, continuations and single characters (bert-base-uncased's special ids)"""
    rng = np.random.default_rng(0)
    vocab = synthetic_vocab(2000)
    words = ["".join(rng.choice(list("abcdefghijklmnopqrstuvwxyz"), int(rng.integers(2, 8)))) for _ in range(300)]
    at = 1400
    for w in words:
        vocab[at] = w
        vocab[at + 300] = "##" + w[: max(1, len(w) // 2)]
        at += 1
    d = tmp_path_factory.mktemp("bert")
    (d / "vocab.txt").write_text("\n".join(vocab) + "\n", encoding="utf-8")
    return d / "vocab.txt", words


def test_tokenizer_matches_transformers_on_the_same_vocabulary(word_vocab):
    from transformers import BertTokenizer as HfBertTokenizer

    path, words = word_vocab
    vocab = {tok: i for i, tok in reversed(list(enumerate(path.read_text(encoding="utf-8").split("\n")[:-1])))}
    hf = HfBertTokenizer(vocab=vocab, do_lower_case=True)
    tk = BertTokenizer.from_file(path, context=100)
    rng = np.random.default_rng(1)
    texts = ["A person riding a horse.", "  two   spaces,\tand a tab\n", "Ärger mit Café-Crème!", "", "don't stop (ever)",
             "中文 mixed in", "x" * 120 + " long", "unknown ✓ glyph", "e-mail: a_b@c.de; 3.14%"] + \
            [" ".join(rng.choice(words, int(rng.integers(1, 20)))) + rng.choice([".", "?", "", " !"]) for _ in range(20)] + \
            ["".join(rng.choice(words, 3)) for _ in range(5)]         # concatenations: whole word + continuations
    mine = tk(texts)
    ref = hf(texts, add_special_tokens=True, max_length=100, padding="max_length", return_tensors="pt").input_ids
    assert mine.shape == (len(texts), 100) and torch.equal(mine, ref)
    assert (mine[:, 0] == 101).all() and int(mine[3, 1]) == 102 and int(mine[3, 2]) == 0   # empty text: [CLS] [SEP] [PAD]
    assert basic_tokens("Hello, World") == ["hello", ",", "world"]
    with pytest.raises(ValueError, match="text_len"):
        tk(" ".join(["a"] * 150))
    import os
    old = os.environ.pop("WISE_AMD_WEIGHTS_DIR", None)
    try:
        with pytest.raises(FileNotFoundError):
            BertTokenizer.default()
        syn = BertTokenizer.default(allow_synthetic=True)             # seeded weights: character-level stand-in
        ids = syn("ab c")[0]
        assert ids[0] == 101 and int((ids != 0).sum()) == 5 and int(ids[4]) == 102
    finally:
        if old is not None:
            os.environ["WISE_AMD_WEIGHTS_DIR"] = old


def test_cnn14_layout_oracle_golden_and_folding():
    lib = _lib.load()
    nb, nf = C.c_int64(), C.c_int64()
    assert lib.wise_cnn14_layout(C.byref(nb), C.byref(nf)) == 0
    sd = random_cnn14_state_dict(0)
    assert list(sd) == [k for k, _ in state_dict_keys()]
    wb, pf = pack_cnn14_weights(sd)
    assert wb.numel() == nb.value and pf.numel() == nf.value and wb.dtype == torch.bfloat16
    convs = 64 * 9 * 64 + sum(CHANNELS[i] * 9 * CHANNELS[i - 1] + CHANNELS[i] * 9 * CHANNELS[i] for i in range(1, 6))
    assert nb.value == convs + 2048 * 2048 + 1024 * 2048 + 1024 * 1024
    assert lib.wise_cnn14_workspace_bytes(1, 192000) > 0 and lib.wise_cnn14_workspace_bytes(1, 31 * 320 - 1) == 0
    # the oracle reproduces its committed vectors
    g = np.load(GOLD / "cnn14.npz")
    w4, w1 = golden_clips()
    taps = {}
    out1 = cnn14_ref.audio_encoder_2022(sd, w1, taps)
    assert np.allclose(out1, g["out1"], atol=2e-6) and np.allclose(taps["lat"].numpy(), g["lat1"], atol=1e-5)
    # BatchNorm folded into the convolution as the packer does it == conv -> BatchNorm (block 2, conv 1)
    from wise_amd.feature.cnn14 import _fold
    import torch.nn.functional as F
    x = torch.randn(1, 64, 9, 6, generator=torch.Generator().manual_seed(1))
    p = "base.conv_block2."
    ref = cnn14_ref._bn(F.conv2d(x, sd[p + "conv1.weight"], padding=1), sd, p + "bn1.")
    wf, shift = _fold(sd, p, "conv1", "bn1")
    w4d = wf.reshape(128, 3, 3, 64).permute(0, 3, 1, 2)      # [Cout, 9*Cin] with k = (kh*3 + kw)*Cin + c -> OIHW
    got = F.conv2d(x, w4d, padding=1) + shift[None, :, None, None]
    assert torch.allclose(got, ref, atol=2e-5)
