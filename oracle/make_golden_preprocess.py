"""TEST INFRASTRUCTURE — renders golden vectors for the image transform with Pillow itself.

Run in the build container (Pillow is installed there):  python -m oracle.make_golden_preprocess
For each geometry the input frame is `np.random.default_rng(seed).integers(0, 256, (3,H,W), uint8)` (regenerated
by the tests from the seed), pushed through what the reference does to it
(/root/reference/src/feature/mlfoundation_openclip.py:81-90 -> torchvision Resize/CenterCrop -> PIL):
    Image.fromarray(HWC) -> .resize((new_w, new_h), BICUBIC) -> .crop(center S x S)
and the resulting uint8 [3,S,S] is stored in tests/golden/preprocess.npz.
"""
from __future__ import annotations

from pathlib import Path

import numpy as np
import PIL
from PIL import Image

from oracle.preprocess_ref import clip_preprocess_u8, resized_geometry

# (H, W, S, seed)
CASES = [
    (240, 320, 224, 11),    # Kinetics-like landscape
    (320, 240, 224, 12),    # portrait
    (1080, 1920, 224, 13),  # 1080p: 4.8x downscale, 21 taps
    (100, 150, 224, 14),    # upscale
    (224, 224, 224, 15),    # nothing to do
    (224, 300, 224, 16),    # crop only
    (333, 517, 224, 17),    # width not a multiple of 4
    (480, 854, 224, 18),    # common unaligned video width
    (2160, 3840, 224, 19),  # 4K: 9.6x downscale
    (360, 640, 336, 20),    # ViT-L/14@336 geometry
]


def pil_reference(frame: np.ndarray, S: int) -> np.ndarray:
    H, W = frame.shape[1:]
    nw, nh, left, top = resized_geometry(H, W, S)
    im = Image.fromarray(np.ascontiguousarray(frame.transpose(1, 2, 0)), mode="RGB")
    if (nw, nh) != (W, H):
        im = im.resize((nw, nh), Image.BICUBIC)
    im = im.crop((left, top, left + S, top + S)).convert("RGB")
    return np.ascontiguousarray(np.asarray(im, dtype=np.uint8).transpose(2, 0, 1))


def case_input(H: int, W: int, seed: int) -> np.ndarray:
    return np.random.default_rng(seed).integers(0, 256, (3, H, W), dtype=np.uint8)


def main():
    out = {}
    for H, W, S, seed in CASES:
        frame = case_input(H, W, seed)
        ref = pil_reference(frame, S)
        mine = clip_preprocess_u8(frame[None], S)[0]
        assert np.array_equal(ref, mine), f"oracle differs from Pillow for {H}x{W}->{S}"
        out[f"out_{H}x{W}_{S}_{seed}"] = ref
    out["cases"] = np.asarray(CASES, dtype=np.int64)
    out["pillow_version"] = np.asarray(PIL.__version__)
    path = Path(__file__).resolve().parents[1] / "tests" / "golden" / "preprocess.npz"
    np.savez_compressed(path, **out)
    print("wrote", path, path.stat().st_size, "bytes; oracle == Pillow", PIL.__version__, "on", len(CASES), "cases")


if __name__ == "__main__":
    main()
