"""TEST INFRASTRUCTURE — CPU oracle for SURVEY.md §8 f2 (GPU preprocess).  Never imported by wise_amd/.

Restates, in numpy integer arithmetic, what the reference does to one decoded frame before the image tower
(/root/reference/src/feature/mlfoundation_openclip.py:81-90):

    uint8 [3,H,W] --F.to_pil_image--> PIL RGB --open_clip eval transform--> fp32 [3,S,S]
    transform = Resize(S, BICUBIC, shorter side) -> CenterCrop(S) -> RGB -> ToTensor -> Normalize(mean, std)

The arithmetic lives in third-party code that the reference calls and that is NOT vendored in it:
  * torchvision==0.17.2 (torch-faiss-requirements.txt:4): Resize -> `_compute_resized_output_size`
    (new_long = int(S * long / short)), `img.resize((w, h), BICUBIC)`; CenterCrop -> top/left =
    int(round((size - S) / 2.0)) (Python round: half to even);
  * Pillow (unpinned by the reference; 12.2.0 in this image) `Image.resize` for 8-bit images = libImaging
    Resample.c: `precompute_coeffs` (double), `normalize_coeffs_8bpc` (22-bit fixed point),
    horizontal pass then vertical pass, each rounding to uint8 through `clip8`.
Pillow IS installed here, so the restatement is pinned bit-for-bit against `PIL.Image.resize` itself
(oracle/make_golden_preprocess.py, tests/test_oracle_cpu.py) — this path's parity is pinned.
"""
from __future__ import annotations

import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2  # Resample.c: coefficients are 22-bit fixed point for 8-bit channels

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def bicubic_filter(x: float) -> float:
    """Resample.c bicubic_filter, a = -0.5 (Keys), support 2."""
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def precompute_coeffs(in_size: int, out_size: int):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for box (0, in_size).
    Returns (ksize, bounds int32 [out,2] = (first input index, count), coeffs int32 [out,ksize])."""
    support_unit = 2.0
    scale = float(in_size) / out_size
    filterscale = scale if scale >= 1.0 else 1.0
    support = support_unit * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)  # C (int) cast: truncation toward zero
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [bicubic_filter((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            # normalize_coeffs_8bpc: round half away from zero, then truncate
            if v < 0:
                kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS))
            else:
                kk[xx, x] = int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return ksize, bounds, kk


def _clip8(acc: np.ndarray) -> np.ndarray:
    """Resample.c clip8: arithmetic shift by PRECISION_BITS, then clamp to 0..255."""
    return np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)


def _pass_last_axis(img: np.ndarray, out_size: int) -> np.ndarray:
    """One resampling pass along the last axis of a uint8 array (ImagingResampleHorizontal_8bpc)."""
    in_size = img.shape[-1]
    ksize, bounds, kk = precompute_coeffs(in_size, out_size)
    out = np.empty(img.shape[:-1] + (out_size,), dtype=np.uint8)
    src = img.astype(np.int64)
    for xx in range(out_size):
        x0, n = int(bounds[xx, 0]), int(bounds[xx, 1])
        acc = (src[..., x0:x0 + n] * kk[xx, :n].astype(np.int64)).sum(axis=-1) + (1 << (PRECISION_BITS - 1))
        # Resample.c accumulates in a 32-bit int; the sums stay inside it (|coeff sums| < 2^23)
        assert np.all(np.abs(acc) < 2 ** 31)
        out[..., xx] = _clip8(acc)
    return out


def pil_resize_bicubic_u8(img: np.ndarray, out_w: int, out_h: int) -> np.ndarray:
    """PIL `Image.resize((out_w, out_h), BICUBIC)` on uint8 [..., H, W] planes: horizontal pass (skipped
    when the width is unchanged), then vertical pass (skipped when the height is unchanged)."""
    H, W = img.shape[-2:]
    x = img
    if out_w != W:
        x = _pass_last_axis(x, out_w)
    if out_h != H:
        x = np.swapaxes(_pass_last_axis(np.swapaxes(x, -1, -2), out_h), -1, -2)
    return np.ascontiguousarray(x)


def resized_geometry(H: int, W: int, S: int):
    """torchvision Resize(S) + CenterCrop(S): (new_w, new_h, left, top)."""
    if W <= H:
        nw, nh = S, int(S * H / W)
    else:
        nw, nh = int(S * W / H), S
    left = int(round((nw - S) / 2.0))
    top = int(round((nh - S) / 2.0))
    return nw, nh, left, top


def clip_preprocess_u8(frames: np.ndarray, S: int) -> np.ndarray:
    """uint8 [n,3,H,W] -> uint8 [n,3,S,S]: resize (shorter side to S) + centre crop, before ToTensor."""
    n, c, H, W = frames.shape
    nw, nh, left, top = resized_geometry(H, W, S)
    r = pil_resize_bicubic_u8(frames, nw, nh)
    return np.ascontiguousarray(r[:, :, top:top + S, left:left + S])


def squash_preprocess_u8(frames: np.ndarray, S: int) -> np.ndarray:
    """uint8 [n,3,H,W] -> uint8 [n,3,S,S]: open_clip resize_mode 'squash' (the SigLIP models' preprocess_cfg) —
    Resize((S, S), BICUBIC) without regard to the aspect ratio, no crop."""
    return pil_resize_bicubic_u8(frames, S, S)


def clip_preprocess(frames: np.ndarray, S: int) -> np.ndarray:
    """Full transform: uint8 [n,3,H,W] -> fp32 [n,3,S,S] = (u8/255 - mean)/std in fp32 (ToTensor, Normalize)."""
    u = clip_preprocess_u8(frames, S).astype(np.float32) / np.float32(255.0)
    mean = np.asarray(CLIP_MEAN, dtype=np.float32).reshape(1, 3, 1, 1)
    std = np.asarray(CLIP_STD, dtype=np.float32).reshape(1, 3, 1, 1)
    return (u - mean) / std
