"""-m gpu: HP-1 audio parity for MS-CLAP '2022' — the Cnn14 HIP kernels (through the C ABI) against the fp32 CPU oracle
(oracle/cnn14_ref.py) and the committed golden vectors.  Tolerance (BASELINE.json north_star): cosine within 1e-3;
activations travel as bf16 between the twelve convolutions (fp32 accumulation)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import cnn14_ref
from oracle.make_golden_cnn14 import golden_clips
from wise_amd.feature.cnn14 import Cnn14Engine, conv3x3_relu, pack_cnn14_weights, random_cnn14_state_dict
from wise_amd.feature.feature_extractor_factory import FeatureExtractorFactory

pytestmark = pytest.mark.gpu


def cosine(a, b):
    a = a.double(); b = b.double()
    return ((a * b).sum(-1) / (a.norm(dim=-1) * b.norm(dim=-1))).min().item()


def rel_l2(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm()).item()


@pytest.fixture(scope="module")
def engine():
    return Cnn14Engine(random_cnn14_state_dict(0), max_batch=4, max_samples=192000)


@pytest.mark.parametrize("B,T,Fq,cin,cout,pool", [
    (1, 8, 8, 64, 64, False), (3, 7, 4, 64, 128, False), (2, 21, 2, 128, 128, False), (1, 5, 3, 256, 512, False),
    (2, 33, 16, 64, 64, False), (1, 1, 1, 128, 256, False),
    (1, 8, 8, 64, 64, True), (3, 7, 5, 64, 128, True), (2, 21, 2, 128, 128, True), (2, 33, 16, 64, 64, True),
    (1, 2, 2, 128, 256, True),
    (1, 331, 128, 64, 256, False), (1, 331, 128, 64, 256, True),      # >= 160 tiles of 256 x 256: the ping-pong kernels
    (2, 93, 4, 128, 512, True),
    (2, 257, 256, 64, 64, False), (2, 257, 256, 64, 64, True)])        # 64 channels, many cells: the 256 x 64 tile
def test_conv3x3_building_block(B, T, Fq, cin, cout, pool):
    """the implicit-GEMM convolution alone against torch's conv2d on the same bf16-rounded operands: edges (zero
    padding at every border, rows past the last tile), both tile widths (64 and 128 output channels), the ping-pong
    tile, and the fused 2x2 average pooling (floor: odd leftovers dropped) in each of them"""
    g = torch.Generator().manual_seed(B * 1000 + T * 10 + cin)
    x = torch.randn(B, T, Fq, cin, generator=g).to(torch.bfloat16)
    w = (torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5).to(torch.bfloat16)
    bias = 0.2 * torch.randn(cout, generator=g)
    ref = F.relu(F.conv2d(x.float().permute(0, 3, 1, 2), w.float(), padding=1) + bias[None, :, None, None])
    if pool:
        ref = F.avg_pool2d(ref, kernel_size=2)
    ref = ref.permute(0, 2, 3, 1)                                     # [B, T', F', Cout]
    wt = w.permute(0, 2, 3, 1).contiguous().reshape(cout, 9 * cin)    # k = (kh*3 + kw)*cin + c
    got = conv3x3_relu(x.cuda(), wt.cuda(), bias.cuda(), pool=pool).float().cpu()
    assert got.shape == ref.shape
    # bf16 output rounding (2^-9 relative) + fp32 summation order
    assert torch.allclose(got, ref, rtol=8e-3, atol=2e-3), (got - ref).abs().max()


def test_frontend_logmel_bn_2022(engine, golden_dir):
    """the shared STFT / mel kernel with the 2022 filterbank (fmax 14000: bands up to 35 FFT bins wide)"""
    w4, _ = golden_clips()
    sd = random_cnn14_state_dict(0)
    engine.forward(w4)
    got = engine.tap(0).cpu()
    taps = {}
    cnn14_ref.cnn14_embedding(sd, w4, taps)
    ref = taps["melbn"]
    g = np.load(golden_dir / "cnn14.npz")
    assert np.allclose(ref[:, :8].numpy(), g["mel_head"], atol=1e-4)
    assert got.shape == ref.shape
    # the tone clip has near-silent bands (dB far below the noise clip's): compare where the band carries energy
    assert (got - ref).abs().max().item() <= 5e-2
    assert cosine(got.reshape(-1, 64), ref.reshape(-1, 64)) >= 1 - 1e-5


def test_golden_4s_clips(engine, golden_dir):
    w4, _ = golden_clips()
    g = np.load(golden_dir / "cnn14.npz")
    out = engine.forward(w4).cpu()
    assert out.shape == (2, 1024) and out.dtype == torch.float32
    assert torch.allclose(out.norm(dim=1), torch.ones(2), atol=1e-5)
    ref = torch.from_numpy(g["out"])
    assert cosine(out, ref) >= 1 - 1e-3
    # the two clips' embeddings are close to each other (cosine 0.96): the DIFFERENCE must point the right way too
    assert cosine((out[0] - out[1])[None], (ref[0] - ref[1])[None]) >= 0.99
    lat = engine.tap(1).float().cpu()
    emb = engine.tap(2).float().cpu()
    assert rel_l2(lat, torch.from_numpy(g["lat"])) <= 2e-2
    assert rel_l2(emb, torch.from_numpy(g["emb"])) <= 2e-2


def test_odd_sizes_batch_independence_and_pipelining(engine, golden_dir):
    w4, w1 = golden_clips()
    g = np.load(golden_dir / "cnn14.npz")
    out1 = engine.forward(w1).cpu()                      # 104 frames: 104 -> 52 -> 26 -> 13 -> 6 -> 3
    assert cosine(out1, torch.from_numpy(g["out1"])) >= 1 - 1e-3
    assert rel_l2(engine.tap(1).float().cpu(), torch.from_numpy(g["lat1"])) <= 2e-2
    a = engine.forward(w4)
    assert torch.equal(engine.forward(w4[1:2]), a[1:2])  # a clip's embedding does not depend on its batch
    c = engine.forward(torch.cat([w4, w4[:1]], dim=0))
    assert torch.equal(c[:2], a) and torch.equal(c[2], a[0])
    assert torch.equal(engine.forward_pipelined(w4).result(), a)
    assert torch.equal(engine.forward_pipelined(w4).result(), a)
    with pytest.raises(ValueError):
        engine.forward(torch.zeros(1, 31 * 320 - 1))      # 31 frames: the fifth pooling would leave nothing


def test_batch_of_10s_clips_against_the_oracle():
    """16 clips x 480000 samples (10 s @48 kHz, the clip length of BASELINE cfg-5): one clip checked against the oracle
    (a few seconds of CPU), rows unit-norm, batch independence bit for bit"""
    B, N = 16, 480000
    sd = random_cnn14_state_dict(0)
    eng = Cnn14Engine(sd, max_batch=B, max_samples=N)
    w = 0.1 * torch.randn(B, N, device="cuda", generator=torch.Generator("cuda").manual_seed(5))
    t = torch.arange(N, device="cuda") / 48000.0
    w[3] = 0.3 * torch.sin(2 * torch.pi * (300.0 * t + 400.0 * t * t)) * (torch.sin(2 * torch.pi * 2.0 * t) > 0)
    out = eng.forward(w)
    assert out.shape == (B, 1024)
    assert torch.allclose(out.norm(dim=1).cpu(), torch.ones(B), atol=1e-5)
    ref = torch.from_numpy(cnn14_ref.audio_encoder_2022(sd, w[3:4].cpu()))
    assert cosine(out[3:4].cpu(), ref) >= 1 - 1e-3
    for b in (0, 3, 15):
        assert torch.equal(eng.forward(w[b:b + 1]), out[b:b + 1])
    assert torch.equal(eng.forward(w[5:12]), out[5:12])


def test_microsoft_clap_2022_plugin_surface(golden_dir):
    """Through the reference's plugin API (microsoft_clap.py:33-51) with the 2022 version token"""
    w4, _ = golden_clips()
    fx = FeatureExtractorFactory("microsoft/clap/2022/seeded-0")
    stereo = torch.stack([w4[1], w4[1]], dim=0)       # [2, N] -> mono mix == w4[1]
    pre = fx.preprocess_audio(stereo)
    assert pre.shape == (1, 1, 192000)
    feats = fx.extract_audio_features(pre)
    assert isinstance(feats, np.ndarray) and feats.shape == (1, 1024) and feats.dtype == np.float32
    g = np.load(golden_dir / "cnn14.npz")
    assert cosine(torch.from_numpy(feats), torch.from_numpy(g["out"][1:2])) >= 1 - 1e-3
    # the reference's own test feeds 408700 samples and checks the dim only (test_feature_extractor.py:37-41)
    long = fx.extract_audio_features(fx.preprocess_audio(torch.rand(1, 408700)))
    assert long.shape == (1, 1024)
    h = fx.extract_audio_features_async(pre)
    assert np.array_equal(h.result(), feats)
