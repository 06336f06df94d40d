"""-m gpu: text tower parity — wise_text_forward through the C ABI against the fp32 CPU oracle
(pinned to transformers' CLIPTextModelWithProjection) and the committed golden vectors.
Tolerance (BASELINE.json north_star): cosine within 1e-3 of the fp32 path."""
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import text_ref
from oracle.make_golden_text import TINY, TINY_GELU, seeded_tokens
from wise_amd import _lib
from wise_amd.feature.text import EOT_TOKEN, SOT_TOKEN, TextEngine, random_text_state_dict, text_spec_for

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).parent / "golden"
COS_TOL = 1e-3


def cosine(a, b):
    a = a.double(); b = b.double()
    return ((a * b).sum(-1) / (a.norm(dim=-1) * b.norm(dim=-1))).min().item()


@pytest.mark.parametrize("B,T,H", [(1, 1, 2), (3, 5, 2), (2, 64, 8), (4, 77, 8), (2, 77, 12), (1, 128, 4), (5, 65, 2)])
def test_causal_attention(B, T, H):
    lib = _lib.lib()
    g = torch.Generator().manual_seed(B * 1000 + T * 10 + H)
    qkv = (torch.randn(B * T, 3 * H * 64, generator=g) * 1.5).to(torch.bfloat16)
    ref = text_ref.attention_causal_ref(qkv.float(), B, T, H)
    qd = qkv.cuda()
    out = torch.empty(B * T, H * 64, dtype=torch.bfloat16, device="cuda")
    _lib.check(lib.wise_attention_causal_bf16(qd.data_ptr(), B, T, H, out.data_ptr(), _lib.stream_ptr()), "attn")
    torch.cuda.synchronize()
    err = (out.float().cpu() - ref).abs().max().item()
    assert err < 3e-2, err
    # row 0 of every sequence sees only key 0: the output is V[0] exactly (to bf16)
    v0 = qkv.float().reshape(B, T, 3, H * 64)[:, 0, 2]
    assert torch.allclose(out.float().cpu().reshape(B, T, H * 64)[:, 0], v0, atol=1e-2)


@pytest.mark.parametrize("spec,fname", [(TINY, "text_tiny.npz"), (TINY_GELU, "text_tiny_gelu.npz"),
                                        (text_spec_for("ViT-B-32", "openai"), "text_b32.npz"),
                                        (text_spec_for("ViT-L-14", "openai"), "text_l14.npz"),
                                        (text_spec_for("ViT-H-14", "laion2b_s32b_b79k"), "text_h14.npz")])
def test_text_forward_matches_golden(spec, fname):
    gold = np.load(GOLD / fname)
    seed, n, tok_seed = (int(v) for v in gold["meta"][:3])
    tokens = seeded_tokens(n, spec.context, tok_seed, spec.vocab)
    assert np.array_equal(tokens, gold["tokens"])
    eng = TextEngine(spec, random_text_state_dict(spec, seed), max_batch=n)
    out = eng.forward(torch.from_numpy(tokens))
    torch.cuda.synchronize()
    got = out.cpu()
    want = torch.from_numpy(gold["out"])
    assert got.shape == want.shape
    assert abs(got.norm(dim=1) - 1).max() < 1e-5
    assert cosine(got, want) > 1 - COS_TOL, cosine(got, want)
    # residual stream after the last block, first 8 positions of every sequence
    res = eng.residual(n).cpu().reshape(n, spec.context, spec.width)[:, :8]
    rw = torch.from_numpy(gold["resid_last"])
    assert cosine(res.reshape(-1, spec.width), rw.reshape(-1, spec.width)) > 1 - 2e-3


def test_pooling_follows_first_argmax_and_batch_independence():
    spec = TINY
    sd = random_text_state_dict(spec, 3)
    eng = TextEngine(spec, sd, max_batch=8)
    V = spec.vocab
    tok = np.zeros((4, spec.context), dtype=np.int32)
    tok[0, :4] = [V - 2, 5, 6, V - 1]                 # ordinary
    tok[1, :6] = [V - 2, 5, V - 1, 7, V - 1, 9]       # the maximum appears twice: the first one pools
    tok[2, :] = 7; tok[2, 0] = V - 2; tok[2, -1] = V - 1   # full context, end-of-text in the last slot
    tok[3, :2] = [V - 2, V - 1]                       # empty prompt
    t = torch.from_numpy(tok)
    with torch.no_grad():
        want = text_ref.text_forward(sd, t, heads=spec.heads, act=spec.act)
    got = eng.forward(t).cpu()
    assert cosine(got, want) > 1 - COS_TOL
    # causal: what follows the first end-of-text cannot matter
    tok2 = tok.copy(); tok2[1, 3:] = 0
    got2 = eng.forward(torch.from_numpy(tok2)).cpu()
    assert torch.allclose(got[1], got2[1], atol=1e-6)
    # a query's embedding does not depend on its batch mates
    alone = eng.forward(t[2:3]).cpu()
    assert torch.allclose(alone[0], got[2], atol=1e-5)


def test_bad_tokens_raise():
    eng = TextEngine(TINY, random_text_state_dict(TINY, 0), max_batch=2)
    with pytest.raises(ValueError):
        eng.forward(torch.zeros(2, 76, dtype=torch.int32))
    with pytest.raises(ValueError):
        eng.forward(torch.full((1, 77), TINY.vocab, dtype=torch.int32))
    with pytest.raises(ValueError):
        eng.forward(torch.zeros(1, 77, dtype=torch.float32))


def test_extractor_text_features_and_end_to_end_search(tmp_path):
    """The reference's query path end to end (feature_search_index.py:100-114): prompt + text -> tokenizer ->
    text tower -> IndexFlatIP search, through the drop-in classes; the oracle replays it on the CPU."""
    from oracle import ip_topk_ref
    from wise_amd.feature.feature_extractor_factory import FeatureExtractorFactory
    from wise_amd.feature.store.feature_store_factory import FeatureStoreFactory, FeatureStoreType
    from wise_amd.index.search_index_factory import SearchIndexFactory

    fid = "mlfoundations/open_clip/ViT-B-32/seeded-0"
    fx = FeatureExtractorFactory(fid)
    texts = ["This is a photo of a dog", "cat", "people cheering at a football match"]
    feats = fx.extract_text_features(texts)
    assert feats.shape == (3, 512) and feats.dtype == np.float32 and feats.flags["C_CONTIGUOUS"]
    assert np.allclose(np.linalg.norm(feats, axis=1), 1.0, atol=1e-5)
    tokens = fx.preprocess_text(texts)
    assert tokens.shape == (3, 77)
    sd = random_text_state_dict(fx.text_spec, 0)
    with torch.no_grad():
        want = text_ref.text_forward(sd, tokens, heads=fx.text_spec.heads, act=fx.text_spec.act).numpy()
    assert ((feats * want).sum(axis=1)).min() > 1 - COS_TOL

    # index of image embeddings, then a text query through FeatureSearchIndex.search
    fdir, idir = tmp_path / "features", tmp_path / "index"
    fdir.mkdir()
    frames = torch.from_numpy(np.random.default_rng(5).integers(0, 256, (96, 3, 224, 224), dtype=np.uint8))
    X = fx.extract_image_features(frames.cuda())
    st = FeatureStoreFactory.create_store(FeatureStoreType.WEBDATASET, "video", str(fdir))
    st.enable_write(2048, 20 * 1024 * 1024)
    for i in range(X.shape[0]):
        st.add(i + 1, X[i:i + 1])
    st.close()
    si = SearchIndexFactory("video", fid, {"features_dir": fdir, "index_dir": idir})
    si.create_index("IndexFlatIP")
    assert si.load_index("IndexFlatIP") is True
    dist, ids = si.search("video", "dog", topk=5)
    assert dist.shape == (5,) and ids.shape == (5,) and ids.dtype == np.int64
    q = fx.extract_text_features(["This is a photo of a dog"])      # the prompt the reference prepends (:24-28,:110)
    D, I = ip_topk_ref.ip_topk(X, q, 5, ids=np.arange(96, dtype=np.int64) + 1)
    assert np.array_equal(ids, I[0]) and np.allclose(dist, D[0], atol=2e-5)


def test_graph_replay_equals_direct_launch():
    """Small batches run from a captured hipGraph; the result must be the direct launch's, bit for bit, and the
    graph must pick up new token ids on every replay."""
    spec = text_spec_for("ViT-B-32", "openai")
    eng = TextEngine(spec, random_text_state_dict(spec, 0), max_batch=8)
    toks = torch.from_numpy(seeded_tokens(6, spec.context, 41, spec.vocab))
    eng.graph_max_batch = 0
    direct = [eng.forward(toks[i:i + 1]).cpu() for i in range(6)]
    direct2 = eng.forward(toks[:2]).cpu()
    eng.graph_max_batch = 4
    for rep in range(2):
        for i in range(6):
            assert torch.equal(eng.forward(toks[i:i + 1]).cpu(), direct[i])
    assert torch.equal(eng.forward(toks[:2]).cpu(), direct2)
    assert len(eng._graphs) == 2
