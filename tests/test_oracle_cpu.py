"""CPU tests (-m "not gpu"): the oracles reproduce the committed golden vectors; the C and numpy
restatements of the flat IP search agree; the HF-CLIP pin holds on a tiny config."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import ip_topk_ref, vit_ref
from oracle.build import build_oracle
from wise_amd.feature.vit import VitSpec, random_state_dict


def _golden(golden_dir, name):
    g = np.load(golden_dir / name)
    s = [int(v) for v in g["spec"]]
    spec = VitSpec(name, s[0], s[1], s[2], s[3], s[4], s[5], s[6], "quick_gelu" if s[7] == 0 else "gelu")
    frames = np.random.default_rng(int(g["frame_seed"])).integers(0, 256, size=(int(g["n_frames"]), 3, s[0], s[0]),
                                                                  dtype=np.uint8)
    return spec, g, torch.from_numpy(frames)


@pytest.mark.parametrize("name", ["vit_tiny.npz", "vit_tiny_gelu.npz", "vit_tiny_h80.npz", "vit_b32.npz"])
def test_vit_oracle_reproduces_golden(golden_dir, name):
    spec, g, frames = _golden(golden_dir, name)
    sd = random_state_dict(spec, int(g["weight_seed"]))
    taps = []
    with torch.no_grad():
        out = vit_ref.vit_forward(sd, vit_ref.normalize_u8(frames), patch=spec.patch, heads=spec.heads, act=spec.act,
                                  taps=taps)
    assert np.allclose(out.numpy(), g["out"], atol=2e-6)
    assert np.allclose(np.linalg.norm(out.numpy(), axis=1), 1.0, atol=1e-6)
    gt = g["taps"]
    mine = np.stack([t.numpy() if gt.ndim == 4 else t[:, 0, :].numpy() for t in taps])
    assert mine.shape == gt.shape and np.allclose(mine, gt, atol=5e-5)
    # the pin against transformers' CLIP recorded when the fixture was made
    assert float(g["pin_out"]) < 2e-5 and float(g["pin_hidden"]) < 1e-3


def test_vit_oracle_pinned_to_hf_clip_tiny():
    """re-run the pin on the tiny config here (seconds): oracle == transformers CLIPVisionModelWithProjection."""
    from oracle.make_golden import TINY, pin_against_hf

    sd = random_state_dict(TINY, 7)
    x = vit_ref.normalize_u8(torch.from_numpy(np.random.default_rng(11).integers(0, 256, size=(3, 3, 64, 64),
                                                                                dtype=np.uint8)))
    d_out, d_hid = pin_against_hf(TINY, sd, x, 2e-5)
    assert d_out < 2e-5


def test_reference_shape_contract():
    """src/feature/test_feature_extractor.py:33-34: ViT-L-14 -> [8,768]; checked on the architecture table."""
    from wise_amd.feature.vit import spec_for

    s = spec_for("ViT-L-14", "openai")
    assert (s.image_size, s.embed_dim, s.tokens, s.act) == (224, 768, 257, "quick_gelu")
    assert spec_for("ViT-B-32", "laion2b_s34b_b79k").act == "gelu"
    assert spec_for("ViT-B-32").flops_per_frame() == 8_817_623_040
    assert spec_for("ViT-L-14").flops_per_frame() == 162_025_537_536
    # the reference's default feature id (extract-features.py:192): ViT-H/14 image tower, head width 80
    from wise_amd.feature.mlfoundation_openclip import list_pretrained
    assert ("xlm-roberta-large-ViT-H-14", "frozen_laion5b_s13b_b90k") in list_pretrained()
    h = spec_for("xlm-roberta-large-ViT-H-14", "frozen_laion5b_s13b_b90k")
    assert (h.width, h.heads, h.width // h.heads, h.layers, h.mlp, h.embed_dim, h.tokens, h.act) == \
        (1280, 16, 80, 32, 5120, 1024, 257, "gelu")
    assert spec_for("ViT-H-14-quickgelu", "dfn5b").act == "quick_gelu"


def _c_oracle():
    lib = C.CDLL(str(build_oracle()))
    lib.wise_oracle_ip_topk.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                        C.c_int64, C.c_void_p, C.c_void_p]
    lib.wise_oracle_ip_topk.restype = None
    return lib


def test_ip_oracles_agree_and_reproduce_golden(golden_dir):
    g = np.load(golden_dir / "ip_topk.npz")
    X = np.random.default_rng(2).standard_normal((4096, 512), dtype=np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    Q = np.random.default_rng(3).standard_normal((8, 512), dtype=np.float32)
    Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    ids = np.arange(4096, dtype=np.int64) + 1
    lib = _c_oracle()
    for k in (1, 10, 100):
        D, I = ip_topk_ref.ip_topk(X, Q, k, ids=ids)
        assert np.array_equal(I, g[f"I{k}"]) and np.allclose(D, g[f"D{k}"], atol=1e-6)
        Dc = np.empty((8, k), np.float32)
        Ic = np.empty((8, k), np.int64)
        lib.wise_oracle_ip_topk(X.ctypes.data, 4096, 512, Q.ctypes.data, 8, k, ids.ctypes.data, 0, Dc.ctypes.data,
                                Ic.ctypes.data)
        assert np.array_equal(Ic, I) and np.allclose(Dc, D, atol=2e-6)
    # faiss padding and tie order in both restatements
    Dc = np.empty((2, 10), np.float32)
    Ic = np.empty((2, 10), np.int64)
    lib.wise_oracle_ip_topk(X.ctypes.data, 7, 512, Q.ctypes.data, 2, 10, None, 1, Dc.ctypes.data, Ic.ctypes.data)
    assert np.array_equal(Ic, g["Ishort"]) and np.all(Dc[:, 7:] == ip_topk_ref.NEG)
    Xt = X[:1024].copy()
    Xt[[5, 17, 900]] = Q[0]
    Dc = np.empty((2, 5), np.float32)
    Ic = np.empty((2, 5), np.int64)
    lib.wise_oracle_ip_topk(Xt.ctypes.data, 1024, 512, Q.ctypes.data, 2, 5, ids.ctypes.data, 0, Dc.ctypes.data,
                            Ic.ctypes.data)
    assert np.array_equal(Ic, g["Itie"]) and list(Ic[0, :3]) == [6, 18, 901]


def test_merge_oracle_properties():
    rng = np.random.default_rng(0)
    X = rng.standard_normal((3000, 32)).astype(np.float32)
    Q = rng.standard_normal((4, 32)).astype(np.float32)
    D, I = ip_topk_ref.ip_topk(X, Q, 10, id_base=1)
    parts_D, parts_I = [], []
    for lo, hi in ((0, 1000), (1000, 1003), (1003, 3000)):  # ragged shards, one smaller than k
        d, i = ip_topk_ref.ip_topk(X[lo:hi], Q, 10, id_base=lo + 1)
        parts_D.append(d)
        parts_I.append(i)
    Dm, Im = ip_topk_ref.merge_topk(np.stack(parts_D), np.stack(parts_I), 10)
    assert np.array_equal(Im, I) and np.array_equal(Dm, D)  # sharded search == unsharded search
    assert ip_topk_ref.recall_at_k(Im, I) == 1.0


def test_htsat_oracle_reproduces_golden_and_layout(golden_dir):
    """oracle/htsat_ref.py against tests/golden/htsat.npz (made next to the HF ClapAudioModel pin), and the
    packed blobs against the library's layout."""
    import ctypes as C

    from oracle import htsat_ref
    from wise_amd import _lib
    from wise_amd.feature.htsat import pack_htsat_weights, random_htsat_state_dict
    from wise_amd.feature import htsat_frontend as fe

    g = np.load(golden_dir / "htsat.npz")
    assert float(g["pin_latent"]) < 1e-4 and float(g["pin_stft"]) < 1e-4
    sd = random_htsat_state_dict(int(g["weight_seed"]))
    rng = np.random.default_rng(int(g["wave_seed"]))
    wave = torch.from_numpy((0.1 * rng.standard_normal((2, 192000))).astype(np.float32))
    with torch.no_grad():
        mel = htsat_ref.logmel(wave[:1])
        assert mel.shape == (1, 601, 64) and np.allclose(mel[:, :8].numpy(), g["mel_head"][:1], atol=1e-4)
        out = htsat_ref.htsat_forward(sd, wave[:1])
    assert np.allclose(out.numpy(), g["out"][:1], atol=2e-5)
    # product-side tables equal the oracle's restatement
    assert np.array_equal(fe.mel_filterbank(), htsat_ref.mel_filterbank())
    start, length, w = fe.sparse_mel()
    dense = np.zeros((64, 513), np.float32)
    for b in range(64):
        dense[b, start[b]: start[b] + length[b]] = w[b, : length[b]]
    assert np.array_equal(dense, fe.mel_filterbank())
    assert torch.equal(fe.rel_pos_index(), htsat_ref.rel_pos_index())
    lib = _lib.load()
    nb, nf = C.c_int64(), C.c_int64()
    assert lib.wise_htsat_layout(C.byref(nb), C.byref(nf)) == 0
    wb, pf = pack_htsat_weights(sd)
    assert (wb.numel(), pf.numel()) == (nb.value, nf.value)
    assert lib.wise_htsat_workspace_bytes(128, 480000) > 0 and lib.wise_htsat_workspace_bytes(1, 100) == 0
