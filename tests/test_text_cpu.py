"""Text tower without a GPU: the oracle against its committed golden vectors, the causal
property, and the host-side weight layout against what the library reports."""
import ctypes as C
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import text_ref
from oracle.make_golden_text import TINY, TINY_GELU, seeded_tokens
from wise_amd import _lib
from wise_amd.feature.text import (EOT_TOKEN, SOT_TOKEN, pack_text_weights, random_text_state_dict, text_spec_for,
                                   text_state_dict_keys)

GOLD = Path(__file__).parent / "golden"


@pytest.mark.parametrize("spec,fname", [(TINY, "text_tiny.npz"), (TINY_GELU, "text_tiny_gelu.npz"),
                                        (text_spec_for("ViT-B-32", "openai"), "text_b32.npz")])
def test_oracle_reproduces_golden(spec, fname):
    gold = np.load(GOLD / fname)
    seed, n, tok_seed = (int(v) for v in gold["meta"][:3])
    tokens = torch.from_numpy(seeded_tokens(n, spec.context, tok_seed, spec.vocab))
    assert np.array_equal(tokens.numpy(), gold["tokens"])
    sd = random_text_state_dict(spec, seed)
    with torch.no_grad():
        out = text_ref.text_forward(sd, tokens, heads=spec.heads, act=spec.act)
    assert np.allclose(out.numpy(), gold["out"], atol=2e-6)
    assert np.allclose(out.norm(dim=1).numpy(), 1.0, atol=1e-6)


def test_oracle_is_causal_and_pools_at_first_argmax():
    spec = TINY
    sd = random_text_state_dict(spec, 2)
    V = spec.vocab
    a = torch.zeros(1, 77, dtype=torch.int64)
    a[0, :5] = torch.tensor([V - 2, 10, 11, V - 1, 0])
    b = a.clone()
    b[0, 4:20] = torch.arange(20, 36)          # junk after the end-of-text token
    with torch.no_grad():
        ea = text_ref.text_forward(sd, a, heads=spec.heads)
        eb = text_ref.text_forward(sd, b, heads=spec.heads)
    assert torch.allclose(ea, eb, atol=1e-6)
    c = a.clone()
    c[0, 1] = 12                               # a change before it does matter
    with torch.no_grad():
        ec = text_ref.text_forward(sd, c, heads=spec.heads)
    assert (ea - ec).abs().max() > 1e-3


def test_weight_layout_matches_library():
    lib = _lib.load()  # host-only entry points: no GPU needed
    for spec in (TINY, TINY_GELU, text_spec_for("ViT-B-32", "openai"), text_spec_for("ViT-L-14", "laion2b_s32b_b82k")):
        cfg = spec.c_config()
        nb, nf = C.c_int64(), C.c_int64()
        assert lib.wise_text_layout(C.byref(cfg), C.byref(nb), C.byref(nf)) == 0
        W, F, L, D = spec.width, spec.mlp, spec.layers, spec.embed_dim
        assert nb.value == L * (4 * W * W + 2 * F * W) + D * W
        assert nf.value == spec.vocab * W + spec.context * W + L * (9 * W + F) + 2 * W
        assert lib.wise_text_workspace_bytes(C.byref(cfg), 1) > 0
    sd = random_text_state_dict(TINY, 0)
    assert list(sd) == [k for k, _ in text_state_dict_keys(TINY)]
    wb, pf = pack_text_weights(TINY, sd)
    cfg = TINY.c_config()
    nb, nf = C.c_int64(), C.c_int64()
    lib.wise_text_layout(C.byref(cfg), C.byref(nb), C.byref(nf))
    assert wb.numel() == nb.value and pf.numel() == nf.value and wb.dtype == torch.bfloat16
    assert text_spec_for("ViT-L-14", "laion2b_s32b_b82k").act == "gelu" and text_spec_for("ViT-B-32").act == "quick_gelu"
    bad = _lib.TextConfig(77, 49408, 500, 12, 8, 2000, 512, 0)
    assert lib.wise_text_layout(C.byref(bad), C.byref(nb), C.byref(nf)) != 0
    assert SOT_TOKEN == 49406 and EOT_TOKEN == 49407
