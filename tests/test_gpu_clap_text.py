"""-m gpu: MS-CLAP caption encoder — the text-tower kernels with the GPT-2 switches, through the
C ABI, against the fp32 CPU oracle (GPT-2 body pinned to transformers' GPT2Model) and the committed golden vectors."""
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import clap_text_ref
from oracle.make_golden_clap_text import TINY, seeded_caption_tokens
from wise_amd.feature.clap_text import CAPTION_SPEC, pack_caption_weights, random_caption_state_dict
from wise_amd.feature.text import TextEngine

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).parent / "golden"
COS_TOL = 1e-3


def cosine(a, b):
    a = a.double(); b = b.double()
    return ((a * b).sum(-1) / (a.norm(dim=-1) * b.norm(dim=-1))).min().item()


@pytest.mark.parametrize("spec,fname", [(TINY, "clap_text_tiny.npz"), (CAPTION_SPEC, "clap_text.npz")])
def test_caption_forward_matches_golden(spec, fname):
    gold = np.load(GOLD / fname)
    seed, n, tok_seed, positions = (int(v) for v in gold["meta"])
    tokens = seeded_caption_tokens(n, spec.context, tok_seed, spec.vocab)
    assert np.array_equal(tokens, gold["tokens"])
    sd = random_caption_state_dict(spec, seed, positions)
    eng = TextEngine(spec, sd, max_batch=n, pack=pack_caption_weights)
    got = eng.forward(torch.from_numpy(tokens)).cpu()
    want = torch.from_numpy(gold["out"])
    assert got.shape == want.shape == (n, 1024)
    assert abs(got.norm(dim=1) - 1).max() < 1e-5
    assert cosine(got, want) > 1 - COS_TOL, cosine(got, want)


def test_pooling_is_last_nonzero_and_padding_is_inert():
    spec = TINY
    sd = random_caption_state_dict(spec, 4, 96)
    eng = TextEngine(spec, sd, max_batch=4, pack=pack_caption_weights)
    eng.graph_max_batch = 0
    V = spec.vocab
    tok = np.zeros((3, spec.context), dtype=np.int32)
    tok[0, :4] = [5, 6, 7, V - 1]
    tok[1, :2] = [9, V - 1]
    tok[2, :] = 11; tok[2, -1] = V - 1
    t = torch.from_numpy(tok)
    with torch.no_grad():
        want = clap_text_ref.caption_forward(sd, t, spec.heads)
    got = eng.forward(t).cpu()
    assert cosine(got, want) > 1 - COS_TOL
    alone = eng.forward(t[1:2]).cpu()
    assert torch.allclose(alone[0], got[1], atol=1e-5)


def test_microsoft_clap_text_features_share_the_audio_space():
    """The reference's caption path through the drop-in class (microsoft_clap.py:53-58): 1024-d unit vectors, the
    same dimension as extract_audio_features, reproducible by the oracle from the same ids."""
    from wise_amd.feature.feature_extractor_factory import FeatureExtractorFactory
    fx = FeatureExtractorFactory("microsoft/clap/2023/seeded-0")
    texts = ["this is the sound of rain", "dog barking"]
    feats = fx.extract_text_features(texts)
    assert feats.shape == (2, 1024) and feats.dtype == np.float32
    assert np.allclose(np.linalg.norm(feats, axis=1), 1.0, atol=1e-5)
    tokens = fx.preprocess_text(texts)
    assert tokens.shape == (2, 77)
    sd = random_caption_state_dict(CAPTION_SPEC, 0)
    with torch.no_grad():
        want = clap_text_ref.caption_forward(sd, tokens, CAPTION_SPEC.heads).numpy()
    assert ((feats * want).sum(axis=1)).min() > 1 - COS_TOL
    with pytest.raises(NotImplementedError):     # a captioning model: no embeddings in the reference either
        FeatureExtractorFactory("microsoft/clap/clapcap/seeded-0")
