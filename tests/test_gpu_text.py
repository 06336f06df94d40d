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
                                        (text_spec_for("ViT-L-14", "openai"), "text_l14.npz")])
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
