"""torch.ops.wise_hip.* — the custom-operator layer over the C ABI (north_star).
CPU: the operators are registered with their schemas, have no CPU kernel (no fallback), and infer shapes on meta tensors.
GPU: each operator returns what the engine / index classes return for the same inputs."""
import numpy as np
import pytest
import torch

import wise_amd.torch_ops as wops


def test_operators_are_registered_with_schemas():
    ops = torch.ops.wise_hip
    for name in wops.SCHEMAS:
        op = getattr(ops, name)
        assert op.default._schema.name == f"wise_hip::{name}"
    s = str(torch.ops.wise_hip.ip_topk.default._schema)
    assert "Tensor X" in s and "int k" in s and "Tensor? ids" in s
    assert wops.register() is wops.register()          # idempotent


def test_no_cpu_kernel_behind_the_operators():
    """A CPU tensor never reaches a kernel: the dispatcher has nothing registered for it (no fallback path)."""
    X = torch.randn(100, 64)
    Q = torch.randn(2, 64)
    with pytest.raises((NotImplementedError, RuntimeError)):
        torch.ops.wise_hip.ip_topk(X, Q, 5, None, 0)
    with pytest.raises((NotImplementedError, RuntimeError)):
        torch.ops.wise_hip.htsat_forward(torch.randn(1, 48000), torch.zeros(4, dtype=torch.bfloat16), torch.zeros(4))


def test_shape_inference_on_meta_tensors():
    X = torch.empty(1000, 512, device="meta")
    Q = torch.empty(3, 512, device="meta")
    D, I = torch.ops.wise_hip.ip_topk(X, Q, 10, None, 1)
    assert D.shape == (3, 10) and D.dtype == torch.float32 and I.dtype == torch.int64 and D.device.type == "meta"
    Xb, norms = torch.ops.wise_hip.ip_shadow_bf16(X)
    assert Xb.shape == X.shape and Xb.dtype == torch.int16 and norms.shape == (2,)
    S = torch.ops.wise_hip.ip_scores(X, Q)
    assert S.shape == (3, 1000)
    im = torch.empty(7, 3, 224, 224, device="meta")
    e = torch.ops.wise_hip.vit_forward(im, torch.empty(8, device="meta"), torch.empty(8, device="meta"),
                                       [224, 32, 768, 12, 12, 3072, 512, 0])
    assert e.shape == (7, 512)
    w = torch.ops.wise_hip.htsat_forward(torch.empty(5, 480000, device="meta"), torch.empty(8, device="meta"),
                                         torch.empty(8, device="meta"))
    assert w.shape == (5, 1024)
    w = torch.ops.wise_hip.cnn14_forward(torch.empty(5, 480000, device="meta"), torch.empty(8, device="meta"),
                                         torch.empty(8, device="meta"))
    assert w.shape == (5, 1024)
    y = torch.ops.wise_hip.conv3x3_relu(torch.empty(2, 9, 6, 64, dtype=torch.bfloat16, device="meta"),
                                        torch.empty(128, 576, dtype=torch.bfloat16, device="meta"),
                                        torch.empty(128, device="meta"), True)
    assert y.shape == (2, 4, 3, 128) and y.dtype == torch.bfloat16
    c = torch.ops.wise_hip.clip_preprocess_u8(torch.empty(4, 3, 240, 320, dtype=torch.uint8, device="meta"), 224)
    assert c.shape == (4, 3, 224, 224) and c.dtype == torch.uint8


@pytest.mark.gpu
def test_search_operators_equal_the_index_classes():
    from wise_amd.index.flat_ip import FlatIPIndex
    g = torch.Generator(device="cuda").manual_seed(3)
    N, d, k = 300000, 512, 10
    X = torch.nn.functional.normalize(torch.randn(N, d, device="cuda", generator=g), dim=1)
    Q = torch.nn.functional.normalize(torch.randn(5, d, device="cuda", generator=g), dim=1)
    ids = torch.arange(N, device="cuda", dtype=torch.int64) * 2 + 7
    ref = FlatIPIndex(d, shadow=False).adopt(X, ids)
    Dr, Ir = ref.search_device(Q, k)
    D, I = torch.ops.wise_hip.ip_topk(X, Q, k, ids, 0)
    assert torch.equal(I, Ir) and torch.equal(D, Dr)
    Xb, norms = torch.ops.wise_hip.ip_shadow_bf16(X)
    counters = torch.zeros(2, dtype=torch.int32, device="cuda")
    for q in range(2):       # one query at a time: the threshold form over the bf16 shadow
        D1, I1 = torch.ops.wise_hip.ip_topk_shadow(X, Xb, norms, Q[q:q + 1], k, ids, 0, counters)
        assert torch.equal(I1, Ir[q:q + 1]) and torch.allclose(D1, Dr[q:q + 1], atol=2e-6)
    assert counters.tolist() == [2, 0]
    Xq, scales, norms8 = torch.ops.wise_hip.ip_shadow_i8(X)
    for q in range(2):       # ... and over the int8 shadow
        D1, I1 = torch.ops.wise_hip.ip_topk_shadow8(X, Xq, scales, norms8, Q[q:q + 1], k, ids, 0, counters)
        assert torch.equal(I1, Ir[q:q + 1]) and torch.allclose(D1, Dr[q:q + 1], atol=2e-6)
    assert counters.tolist() == [4, 0]
    # wrong dtypes or a strided view are refused, not read as garbage
    with pytest.raises(ValueError):
        torch.ops.wise_hip.ip_topk(X, Q, k, ids.to(torch.int32), 0)
    with pytest.raises(ValueError):
        torch.ops.wise_hip.ip_topk_shadow(X, Xb[:, ::2], norms, Q[:1], k, ids, 0, counters)
    with pytest.raises(ValueError):
        torch.ops.wise_hip.ip_topk_shadow8(X, Xq.to(torch.int16), scales, norms8, Q[:1], k, ids, 0, counters)
    # two half shards merged == the whole
    h = N // 2
    Da, Ia = torch.ops.wise_hip.ip_topk(X[:h], Q, k, ids[:h], 0)
    Db, Ib = torch.ops.wise_hip.ip_topk(X[h:], Q, k, ids[h:], 0)
    Dm, Im = torch.ops.wise_hip.topk_merge(torch.stack([Da, Db]), torch.stack([Ia, Ib]), k)
    assert torch.equal(Im, Ir) and torch.allclose(Dm, Dr, atol=2e-6)
    rows = torch.ops.wise_hip.reconstruct_batch(X, ids, 0, ids[[5, 77]])
    assert torch.equal(rows, X[[5, 77]])
    S = torch.ops.wise_hip.ip_scores(X[:5000], Q)
    sel = torch.ops.wise_hip.select_topk(S, 64)
    want = torch.topk(S, 64, dim=1).indices.sort(dim=1).values
    assert torch.equal(sel, want)


@pytest.mark.gpu
def test_feature_operators_equal_the_engines(golden_dir):
    from wise_amd.feature.htsat import HtsatEngine, random_htsat_state_dict
    from wise_amd.feature.text import TextEngine, random_text_state_dict, text_spec_for
    from wise_amd.feature.vit import VitEngine, VitSpec, random_state_dict

    spec = VitSpec("t", 64, 32, 128, 2, 2, 512, 64, "quick_gelu")
    eng = VitEngine(spec, random_state_dict(spec, 7), max_batch=5)
    frames = torch.from_numpy(np.random.default_rng(1).integers(0, 256, size=(5, 3, 64, 64), dtype=np.uint8)).cuda()
    cfg = [spec.image_size, spec.patch, spec.width, spec.layers, spec.heads, spec.mlp, spec.embed_dim, 0]
    assert torch.equal(torch.ops.wise_hip.vit_forward(frames, eng.wb, eng.pf, cfg), eng.forward(frames))
    # (the operator takes the blobs in wise_htsat_layout()'s plain order = wise_htsat_forward; the engine's default packs the
    #  stage-2 / -3 MLP weights as wise_mlp_stream's stream for wise_htsat_forward2 flags bit 1)
    heng = HtsatEngine(random_htsat_state_dict(0), max_batch=2, max_samples=192000, ln_fold=False, mlp_stream=False, attn_stream=False)
    w = 0.1 * torch.randn(2, 192000, device="cuda", generator=torch.Generator("cuda").manual_seed(2))
    assert torch.equal(torch.ops.wise_hip.htsat_forward(w, heng.wb, heng.pf), heng.forward(w))
    dflt = HtsatEngine(random_htsat_state_dict(0), max_batch=2, max_samples=192000).forward(w)
    assert float((1 - (dflt.double() * heng.forward(w).double()).sum(1)).max()) <= 1e-4
    from wise_amd.feature.cnn14 import Cnn14Engine, conv3x3_relu, random_cnn14_state_dict
    ceng = Cnn14Engine(random_cnn14_state_dict(0), max_batch=2, max_samples=192000)
    assert torch.equal(torch.ops.wise_hip.cnn14_forward(w, ceng.wb, ceng.pf), ceng.forward(w))
    cx = torch.randn(2, 9, 6, 64, device="cuda").to(torch.bfloat16)
    cw = (0.05 * torch.randn(128, 576, device="cuda")).to(torch.bfloat16)
    cbias = torch.randn(128, device="cuda")
    assert torch.equal(torch.ops.wise_hip.conv3x3_relu(cx, cw, cbias, True), conv3x3_relu(cx, cw, cbias, True))
    assert torch.ops.wise_hip.conv3x3_relu(cx, cw, cbias, False).shape == (2, 9, 6, 128)
    tspec = text_spec_for("ViT-B-32", "openai")
    teng = TextEngine(tspec, random_text_state_dict(tspec, 0), max_batch=4)
    toks = torch.zeros(3, tspec.context, dtype=torch.int32, device="cuda")
    toks[:, 0] = 49406
    toks[:, 1:4] = torch.tensor([[320, 1125, 539], [5, 6, 7], [9, 1, 2]], dtype=torch.int32)
    toks[:, 4] = 49407
    c = teng.cfg
    tcfg = [c.context, c.vocab, c.width, c.layers, c.heads, c.mlp, c.embed_dim, c.act, c.pool, c.head]
    assert torch.equal(torch.ops.wise_hip.text_forward(toks, teng.wb, teng.pf, tcfg), teng.forward(toks))
    from oracle.make_golden_xlmr import TINY as XT, seeded_tokens as xtok
    from wise_amd.feature.xlmr_text import XlmrTextEngine, random_xlmr_state_dict
    xeng = XlmrTextEngine(XT, random_xlmr_state_dict(XT, 1), max_batch=4)
    xt = torch.from_numpy(xtok(3, XT, 4)).cuda()
    xc = xeng.cfg
    xcfg = [xc.context, xc.vocab, xc.max_positions, xc.width, xc.layers, xc.heads, xc.mlp, xc.proj_hidden, xc.embed_dim,
            xc.pad_id]
    assert torch.equal(torch.ops.wise_hip.xlmr_forward(xt, xeng.wb, xeng.pf, xcfg), xeng.forward(xt))
    raw = torch.from_numpy(np.random.default_rng(3).integers(0, 256, size=(2, 3, 120, 160), dtype=np.uint8)).cuda()
    from wise_amd.feature.preprocess import ClipPreprocessor
    assert torch.equal(torch.ops.wise_hip.clip_preprocess_u8(raw, 64), ClipPreprocessor(64)(raw))
