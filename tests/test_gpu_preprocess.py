"""SURVEY.md §8 f2 on the GPU: wise_preproc_u8 (through the C ABI) against Pillow-rendered golden vectors and the
oracle, bit for bit; then the whole uint8 -> embedding path against the reference's CPU PIL loop."""
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import preprocess_ref as ref
from oracle.make_golden_preprocess import CASES, case_input

pytestmark = pytest.mark.gpu
GOLD = np.load(Path(__file__).parent / "golden" / "preprocess.npz")


@pytest.fixture(scope="module")
def pre224():
    from wise_amd.feature.preprocess import ClipPreprocessor
    return ClipPreprocessor(224)


@pytest.mark.parametrize("matrix_cores", [1, 4, False, "auto"])
@pytest.mark.parametrize("H,W,S,seed", CASES)
def test_kernel_reproduces_pillow_golden(H, W, S, seed, matrix_cores):
    """every form of the kernel: the integer matrix-core form with one and with four waves per tile (where the plan offers it:
    moderate scales), the dot-product form (everywhere), and whichever the first call's measurement picks"""
    from wise_amd.feature.preprocess import ClipPreprocessor
    frame = torch.from_numpy(case_input(H, W, seed))[None].cuda()
    out = ClipPreprocessor(S, matrix_cores=matrix_cores)(frame)
    torch.cuda.synchronize()
    got = out.cpu().numpy()[0]
    want = GOLD[f"out_{H}x{W}_{S}_{seed}"]
    assert got.shape == want.shape and np.array_equal(got, want), \
        f"{int((got != want).sum())} bytes differ, max {int(np.abs(got.astype(int) - want).max())}"


@pytest.mark.parametrize("H,W,n", [(240, 320, 19), (270, 481, 5), (720, 1280, 9), (64, 48, 3), (226, 224, 8)])
def test_batches_match_oracle(pre224, H, W, n):
    frames = np.random.default_rng(H * 7 + W).integers(0, 256, (n, 3, H, W), dtype=np.uint8)
    dev = torch.from_numpy(frames).cuda()
    got = pre224(dev)
    torch.cuda.synchronize()
    assert np.array_equal(got.cpu().numpy(), ref.clip_preprocess_u8(frames, 224))


def test_matrix_core_form_is_what_runs_on_video_frames_and_equals_the_dot_product_form():
    """240p ... 1080p frames (and a squashed SigLIP geometry) are offered the matrix-core kernel (plan.reserved bit 1); a 4K frame
    does not fit its LDS budget and stays on the dot-product kernel.  Same bytes from all three, whichever "auto" picks."""
    from wise_amd.feature.preprocess import ClipPreprocessor, make_plan
    for (H, W, S, squash, want_mfma) in [(240, 320, 224, False, True), (480, 854, 224, False, True), (720, 1280, 224, False, True),
                                         (1080, 1920, 224, False, True), (360, 640, 384, True, True), (2160, 3840, 224, False, False)]:
        assert bool(make_plan(H, W, S, squash).reserved & 2) is want_mfma, (H, W, S)
        frames = torch.from_numpy(np.random.default_rng(H + W).integers(0, 256, (3, 3, H, W), dtype=np.uint8)).cuda()
        auto = ClipPreprocessor(S, squash=squash)
        a = auto(frames)
        assert auto.chosen[(H, W)] in ("dot products", "matrix cores, one wave per tile", "matrix cores, four waves per tile")
        assert want_mfma or auto.chosen[(H, W)] == "dot products"
        for mc in (False, 1, 4):
            assert torch.equal(a, ClipPreprocessor(S, squash=squash, matrix_cores=mc)(frames)), (H, W, S, mc)
        assert torch.equal(a, auto(frames))


def test_constant_and_extreme_frames(pre224):
    """Saturation: clip8 must clamp overshoot of the negative bicubic lobes exactly as Pillow does."""
    H, W = 300, 400
    f = np.zeros((4, 3, H, W), dtype=np.uint8)
    f[1] = 255
    f[2, :, ::2, :] = 255                      # horizontal stripes
    f[3, :, :, (np.arange(W) // 3) % 2 == 0] = 255   # vertical bars
    got = pre224(torch.from_numpy(f).cuda()).cpu().numpy()
    want = ref.clip_preprocess_u8(f, 224)
    assert np.array_equal(got, want)
    assert got[0].max() == 0 and got[1].min() == 255


def test_bad_input_raises(pre224):
    with pytest.raises(ValueError):
        pre224(torch.zeros(2, 3, 32, 32, dtype=torch.float32))
    with pytest.raises(ValueError):
        pre224(torch.zeros(3, 32, 32, dtype=torch.uint8))


def test_uint8_frames_to_embeddings_match_the_cpu_pil_path():
    """extract(preprocess_image_device(u8)) == extract(preprocess_image(u8)): the reference's CPU loop
    (mlfoundation_openclip.py:81-101) and the all-GPU path give the same embeddings."""
    from wise_amd.feature.feature_extractor_factory import FeatureExtractorFactory
    fx = FeatureExtractorFactory("mlfoundations/open_clip/ViT-B-32/seeded-0")
    frames = torch.from_numpy(np.random.default_rng(2).integers(0, 256, (8, 3, 240, 320), dtype=np.uint8))
    cpu_path = fx.extract_image_features(fx.preprocess_image(frames))
    dev_u8 = fx.preprocess_image_device(frames.cuda())
    # the uint8 crop is what PIL produced before ToTensor
    pil_u8 = ref.clip_preprocess_u8(frames.numpy(), 224)
    assert np.array_equal(dev_u8.cpu().numpy(), pil_u8)
    gpu_path = fx.extract_image_features(dev_u8)
    cos = (cpu_path * gpu_path).sum(axis=1)
    assert cos.min() > 1 - 1e-4, cos
    with pytest.raises(ValueError):
        fx.preprocess_image_device(frames.float())


@pytest.mark.parametrize("H,W,S,n", [(240, 320, 384, 5), (720, 1280, 384, 3), (97, 211, 224, 4), (384, 384, 384, 2),
                                      (600, 450, 256, 3), (1080, 1920, 384, 2)])
def test_squash_transform_is_bit_exact(H, W, S, n):
    """the SigLIP models' transform (Resize((S, S)), no crop) on the same kernel: equal to the Pillow-pinned oracle"""
    from wise_amd.feature.preprocess import ClipPreprocessor

    frames = np.random.default_rng(H + 3 * W + S).integers(0, 256, (n, 3, H, W), dtype=np.uint8)
    got = ClipPreprocessor(S, squash=True)(torch.from_numpy(frames).cuda())
    torch.cuda.synchronize()
    want = ref.squash_preprocess_u8(frames, S)
    assert np.array_equal(got.cpu().numpy(), want), f"{int((got.cpu().numpy() != want).sum())} bytes differ"
