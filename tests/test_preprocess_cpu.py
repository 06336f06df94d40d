"""SURVEY.md §8 f2 (image transform) without a GPU: the oracle against Pillow-rendered golden vectors and against
Pillow itself, and the library's HOST-side plan/tap tables against the oracle."""
from pathlib import Path

import numpy as np
import pytest

from oracle import preprocess_ref as ref
from oracle.make_golden_preprocess import CASES, case_input
from wise_amd.feature import preprocess as pp

GOLD = np.load(Path(__file__).parent / "golden" / "preprocess.npz")


@pytest.mark.parametrize("H,W,S,seed", CASES)
def test_oracle_reproduces_pillow_golden(H, W, S, seed):
    out = ref.clip_preprocess_u8(case_input(H, W, seed)[None], S)[0]
    assert np.array_equal(out, GOLD[f"out_{H}x{W}_{S}_{seed}"])


def test_oracle_against_live_pillow():
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(5)
    for H, W in [(97, 211), (256, 256), (600, 450), (225, 224)]:
        frame = rng.integers(0, 256, (3, H, W), dtype=np.uint8)
        nw, nh, left, top = ref.resized_geometry(H, W, 224)
        im = Image.fromarray(np.ascontiguousarray(frame.transpose(1, 2, 0)), mode="RGB")
        if (nw, nh) != (W, H):
            im = im.resize((nw, nh), Image.BICUBIC)
        im = im.crop((left, top, left + 224, top + 224))
        want = np.asarray(im).transpose(2, 0, 1)
        assert np.array_equal(ref.clip_preprocess_u8(frame[None], 224)[0], want)


@pytest.mark.parametrize("in_size,out_size", [(320, 298), (240, 224), (1920, 398), (150, 336), (3840, 398),
                                              (517, 347), (1000, 6054), (7, 224), (224, 225), (1081, 224)])
def test_library_taps_match_oracle(in_size, out_size):
    ks, first, count, coef = pp.pillow_taps(in_size, out_size)
    oks, obounds, okk = ref.precompute_coeffs(in_size, out_size)
    assert ks == oks
    assert np.array_equal(first, obounds[:, 0]) and np.array_equal(count, obounds[:, 1])
    assert np.array_equal(coef, okk)


def test_plan_geometry_matches_torchvision_rules():
    rng = np.random.default_rng(9)
    sizes = [(240, 320), (320, 240), (1080, 1920), (225, 224), (224, 227), (229, 224), (333, 517), (100, 150)]
    sizes += [tuple(int(v) for v in rng.integers(64, 1500, 2)) for _ in range(40)]
    for H, W in sizes:
        plan = pp.make_plan(H, W, 224)
        assert (plan.new_w, plan.new_h, plan.left, plan.top) == ref.resized_geometry(H, W, 224), (H, W)
        assert plan.lds_bytes <= 64 * 1024 and plan.tile in (8, 16, 32)
        assert plan.table_bytes == pp.plan_tables(plan).nbytes


def test_plan_tables_hold_the_cropped_taps():
    """The blob is the oracle's taps for the cropped columns/rows, shifted by (first & 3) into whole dwords and
    stored as three byte planes of (tap + 2^22) per group of four taps (the v_dot4_u32_u8 operand form)."""

    def unpack(planes):  # [S, nd*4] int32 = [S][nd]{p2,p1,p0,0} -> taps [S, nd*4]
        q = planes.reshape(planes.shape[0], -1, 4).view(np.uint32)
        assert not q[:, :, 3].any()
        taps = np.zeros(q.shape[:2] + (4,), dtype=np.int64)
        for b in range(4):
            byte = lambda v: ((v >> (8 * b)) & 255).astype(np.int64)
            taps[:, :, b] = (byte(q[:, :, 0]) << 16) + (byte(q[:, :, 1]) << 8) + byte(q[:, :, 2]) - (1 << 22)
        return taps.reshape(planes.shape[0], -1).astype(np.int32)

    H, W, S = 480, 854, 224
    plan = pp.make_plan(H, W, S)
    tab = pp.plan_tables(plan)
    hstart, vstart = tab[:S], tab[S:2 * S]
    hco = unpack(tab[2 * S:2 * S + S * plan.ndh * 4].reshape(S, plan.ndh * 4))
    vco = unpack(tab[2 * S + S * plan.ndh * 4:2 * S + S * (plan.ndh + plan.ndv) * 4].reshape(S, plan.ndv * 4))
    _, hb, hk = ref.precompute_coeffs(W, plan.new_w)
    for x in range(S):
        f, n = hb[x + plan.left]
        assert hstart[x] == f >> 2
        want = np.zeros(plan.ndh * 4, dtype=np.int32)
        want[(f & 3):(f & 3) + n] = hk[x + plan.left, :n]
        assert np.array_equal(hco[x], want)
    # height 480 -> 224 rows
    _, vb, vk = ref.precompute_coeffs(H, plan.new_h)
    for y in range(S):
        f, n = vb[y + plan.top]
        assert vstart[y] == f >> 2
        want = np.zeros(plan.ndv * 4, dtype=np.int32)
        want[(f & 3):(f & 3) + n] = vk[y + plan.top, :n]
        assert np.array_equal(vco[y], want)


def test_plan_rejects_what_the_kernel_cannot_do():
    with pytest.raises(ValueError):
        pp.make_plan(240, 320, 222)        # output edge not a multiple of 4
    with pytest.raises(ValueError):
        pp.make_plan(16384, 16384, 224)    # 73x downscale: a tile's input does not fit LDS
    with pytest.raises(ValueError):
        pp.make_plan(0, 10, 224)


def test_squash_oracle_against_live_pillow_and_plan_geometry():
    """open_clip resize_mode 'squash' (the SigLIP models): Resize((S, S), BICUBIC), no crop — the oracle equals Pillow bit
    for bit, and the library's squash plan has new size S x S with no crop offset"""
    Image = pytest.importorskip("PIL.Image")
    from wise_amd.feature.preprocess import make_plan

    rng = np.random.default_rng(6)
    for H, W, S in [(240, 320, 384), (720, 1280, 384), (97, 211, 224), (384, 384, 384), (600, 450, 256), (384, 500, 384)]:
        frame = rng.integers(0, 256, (3, H, W), dtype=np.uint8)
        im = Image.fromarray(np.ascontiguousarray(frame.transpose(1, 2, 0)), mode="RGB")
        want = np.asarray(im.resize((S, S), Image.BICUBIC) if (W, H) != (S, S) else im).transpose(2, 0, 1)
        assert np.array_equal(ref.squash_preprocess_u8(frame[None], S)[0], want)
        p = make_plan(H, W, S, squash=True)
        assert (p.new_w, p.new_h, p.left, p.top, p.reserved & 1) == (S, S, 0, 0, 1)   # bit 0 = squash (bits 1, 2: matrix-core forms)
    q = make_plan(240, 320, 224)
    assert q.reserved & 1 == 0 and (q.new_w, q.new_h) == (298, 224)
