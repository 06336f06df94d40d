"""-m gpu: MS-CLAP 2022 caption encoder parity (bert-base-uncased + msclap Projection, microsoft_clap.py:53-58) —
wise_xlmr_forward with the BERT switches, through the C ABI, against the fp32 CPU oracle (pinned to transformers'
BertModel) and the committed golden vectors.  Tolerance (BASELINE.json north_star): cosine within 1e-3 of the fp32 path."""
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import clap_bert_ref
from oracle.make_golden_clap_bert import TINY, seeded_tokens
from wise_amd.feature.clap_bert import CLAP_BERT_SPEC, pack_clap_bert_weights, random_clap_bert_state_dict
from wise_amd.feature.feature_extractor_factory import FeatureExtractorFactory
from wise_amd.feature.xlmr_text import XlmrTextEngine

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).parent / "golden"
COS_TOL = 1e-3


def cosine(a, b):
    a = a.double(); b = b.double()
    return ((a * b).sum(-1) / (a.norm(dim=-1) * b.norm(dim=-1))).min().item()


@pytest.mark.parametrize("spec,fname", [(TINY, "clap_bert_tiny.npz"), (CLAP_BERT_SPEC, "clap_bert_base.npz")])
def test_caption_encoder_matches_golden(spec, fname):
    gold = np.load(GOLD / fname)
    seed, n, tok_seed = (int(v) for v in gold["meta"][:3])
    tokens = seeded_tokens(n, spec, tok_seed)
    assert np.array_equal(tokens, gold["tokens"])
    eng = XlmrTextEngine(spec, random_clap_bert_state_dict(spec, seed), max_batch=n, pack=pack_clap_bert_weights)
    out = eng.forward(torch.from_numpy(tokens))
    torch.cuda.synchronize()
    got, want = out.cpu(), torch.from_numpy(gold["out"])
    assert got.shape == want.shape == (n, 1024)
    assert abs(got.norm(dim=1) - 1).max() < 1e-5
    assert cosine(got, want) > 1 - COS_TOL, cosine(got, want)
    # hidden state of the [CLS] row of every sequence after the last layer
    res = eng.residual(n).cpu().reshape(n, spec.context, spec.width)[:, 0]
    assert cosine(res, torch.from_numpy(gold["taps"][-1])) > 1 - 2e-3


def test_padding_batch_sizes_and_position_mode():
    spec = TINY
    sd = random_clap_bert_state_dict(spec, 3)
    eng = XlmrTextEngine(spec, sd, max_batch=8, pack=pack_clap_bert_weights)
    tok = np.zeros((4, spec.context), dtype=np.int32)
    tok[0, :4] = [101, 1005, 1006, 102]
    tok[1, :2] = [101, 102]                               # empty text
    tok[2, :] = 1007; tok[2, 0] = 101; tok[2, -1] = 102   # full context
    tok[3, :9] = [101, 1009, 1008, 1007, 1006, 1005, 1004, 1010, 102]
    t = torch.from_numpy(tok)
    want = clap_bert_ref.caption_forward_2022(sd, t, heads=spec.heads)
    got = eng.forward(t).cpu()
    assert cosine(got, want) > 1 - COS_TOL
    alone = torch.cat([eng.forward(t[i:i + 1]).cpu() for i in range(4)])
    assert cosine(alone, got) > 1 - 1e-5 and (alone - got).abs().max() < 1e-3
    big = eng.forward(t.repeat(70, 1)).cpu()              # 280 rows: more than one 256-row tile of pooled rows
    assert cosine(big[:4], got) > 1 - 1e-5 and torch.equal(big[:4], big[-4:])
    # absolute positions: the same words one slot later are a different text (with RoBERTa's positions they would also
    # differ; what must NOT happen is a dependence on what lies behind the padding)
    tok2 = tok.copy(); tok2[0, 4:20] = 0
    assert torch.equal(eng.forward(torch.from_numpy(tok2)).cpu()[0], got[0])
    bad = tok.copy(); bad[0, 2] = 0
    with pytest.raises(ValueError, match="right-padded"):
        eng.forward(torch.from_numpy(bad))


def test_microsoft_clap_2022_text_and_audio_share_a_space():
    """Through the reference's plugin API (microsoft_clap.py:42-58) with the 2022 version token and seeded weights: text
    and audio land in one 1024-d space, a search over audio embeddings works end to end"""
    fx = FeatureExtractorFactory("microsoft/clap/2022/seeded-0")
    texts = ["a dog barks", "rain on a tin roof, far away"]
    tok = fx.preprocess_text(texts)
    assert tok.shape == (2, 100) and int(tok[0, 0]) == 101 and (tok[:, -1] == 0).all()
    feats = fx.extract_text_features(texts)
    assert isinstance(feats, np.ndarray) and feats.shape == (2, 1024) and feats.dtype == np.float32
    assert np.allclose(np.linalg.norm(feats, axis=1), 1.0, atol=1e-5)
    want = clap_bert_ref.caption_forward_2022(random_clap_bert_state_dict(CLAP_BERT_SPEC, 0), tok).numpy()
    assert ((feats * want).sum(axis=1)).min() > 1 - COS_TOL
    audio = fx.extract_audio_features(fx.preprocess_audio(0.1 * torch.randn(1, 96000)))
    assert audio.shape == (1, 1024) and fx.get_output_dim() == 1024
    from wise_amd.index.flat_ip import FlatIPIndex
    idx = FlatIPIndex(1024)
    idx.add_with_ids(np.concatenate([audio, feats]).astype(np.float32), np.arange(3, dtype=np.int64))
    D, I = idx.search(feats[:1], 2)
    assert int(I[0, 0]) == 1 and abs(float(D[0, 0]) - 1.0) < 1e-4
