"""-m gpu: the LayerNorm fold of the image tower (wise_vit_config.ln_fold, include/wise_hip.h) — open_clip's ln_1 / ln_2 (as
reached from src/feature/mlfoundation_openclip.py:99) carried by the GEMMs around them instead of by LayerNorm launches.

The two GEMM forms through the C ABI against float64 torch references, their independence of the row count (one reduction
tree whatever tile a batch size selects: a frame's embedding must not depend on the batch it sits in), and the tower in fold
mode against the same golden vectors and oracle as the unfolded tower (cosine within 1e-3 of the fp32 path)."""
import numpy as np
import pytest
import torch

from oracle import vit_ref
from tests.test_gpu_vit import COS_TOL, bf16_round, cosine, load_golden
from wise_amd import _lib
from wise_amd.feature.vit import VitEngine, checkpoint_like_state_dict, fold_layernorm, random_state_dict, tile_out_proj

pytestmark = pytest.mark.gpu


def _act(ref, mode):
    if mode == 1:
        return ref * torch.sigmoid(1.702 * ref)
    if mode == 2:
        return 0.5 * ref * (1 + torch.erf(ref / 2 ** 0.5))
    if mode == 5:
        return 0.5 * ref * (1 + torch.tanh(0.7978845608028654 * (ref + 0.044715 * ref ** 3)))
    return ref


def _fold_bf16(A, W, bias, rstd, mode):
    lib = _lib.lib()
    M, K = A.shape
    N = W.shape[0]
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    _lib.check(lib.wise_gemm_fold_bf16(A.data_ptr(), W.data_ptr(), bias.data_ptr(), rstd.data_ptr(), M, N, K, mode,
                                       out.data_ptr(), _lib.stream_ptr()), "wise_gemm_fold_bf16")
    return out


def _counters(stats, M):
    lib = _lib.lib()
    o, n = lib.wise_gemm_fold_counters_offset(M) // 4, lib.wise_gemm_fold_counters_bytes(M) // 4
    return stats.view(torch.int32)[o:o + n]


def hilo(x):
    """fp32 [M,N] -> the residual stream's storage: one bf16 tensor [2,M,N], hi = bf16(x), lo = bf16(x - hi)"""
    hi = x.to(torch.bfloat16)
    return torch.stack([hi, (x - hi.float()).to(torch.bfloat16)])


def _fold_resid(A, W, bias, xs, eps=1e-5, stats=None, group32=0):
    """xs: [2,M,N] bf16 (hi, lo), updated in place.  -> stats"""
    lib = _lib.lib()
    M, K = A.shape
    N = W.shape[0]
    nbytes = lib.wise_gemm_fold_stats_bytes(M, N)
    assert nbytes >= M * 4 + M * (N // 32) * 8 + (M // 128 + 1) * 4 and nbytes % 4 == 0
    if stats is None:
        stats = torch.full((nbytes // 4,), float("nan"), dtype=torch.float32, device="cuda")
        _counters(stats, M).zero_()                               # only the counters must be zero on entry
    assert xs.is_contiguous() and xs.shape == (2, M, N)
    _lib.check(lib.wise_gemm_fold_resid(A.data_ptr(), W.data_ptr(), bias.data_ptr(), M, N, K, xs.data_ptr(), M * N,
                                        stats.data_ptr(), eps, group32, _lib.stream_ptr()), "wise_gemm_fold_resid")
    return stats


@pytest.mark.parametrize("M,N,K", [(12800, 2304, 768), (12800, 3072, 768), (6400, 2304, 768), (2048, 3072, 768), (256, 2304, 768),
                                    (512, 768, 256), (1280, 3072, 1024), (128, 128, 192),
                                    # HTSAT: QKV and fc1 of stages 2 - 4
                                    (131072, 576, 192), (131072, 768, 192), (32768, 1152, 384), (32768, 1536, 384), (8192, 2304, 768)])
@pytest.mark.parametrize("mode", [0, 1, 2, 5])
def test_gemm_fold_consumer(M, N, K, mode):
    if M > 2048 and mode in (2, 5):
        pytest.skip("large shapes: plain and QuickGELU epilogues")
    g = torch.Generator().manual_seed(M + N + K + mode)
    A = bf16_round(torch.randn(M, K, generator=g) * 3.0)           # un-normalised rows: the row scale brings them back
    W = bf16_round(torch.randn(N, K, generator=g) * K ** -0.5)
    bias = torch.randn(N, generator=g)
    rstd = torch.rand(M, generator=g) * 0.3 + 0.2
    ref = _act(rstd.double()[:, None] * (A.double() @ W.double().t()) + bias.double(), mode)
    out = _fold_bf16(A.to(torch.bfloat16).cuda(), W.to(torch.bfloat16).cuda(), bias.cuda(), rstd.cuda(), mode)
    err = (out.float().cpu().double() - ref).abs().max().item()
    assert err <= max(3e-2, float(ref.abs().max()) * 2.0 ** -8), err      # one bf16 rounding of the result


@pytest.mark.parametrize("M,N,K,g32", [(12800, 768, 768, 0), (12800, 768, 3072, 0), (6400, 768, 768, 0), (2048, 768, 3072, 0),
                                        (256, 768, 768, 0), (1280, 1024, 4096, 0), (128, 128, 192, 0), (384, 256, 256, 0),
                                        # MS-CLAP HTSAT's stages (statistics per 32 columns: 96-column wave parts)
                                        (131072, 192, 192, 1), (131072, 192, 768, 1), (32768, 384, 384, 1), (32768, 384, 1536, 1),
                                        (8192, 768, 768, 1), (8192, 768, 3072, 1), (1024, 192, 768, 1), (256, 384, 384, 1)])
def test_gemm_fold_producer(M, N, K, g32):
    g = torch.Generator().manual_seed(M + N + K)
    A = bf16_round(torch.randn(M, K, generator=g))
    W = bf16_round(torch.randn(N, K, generator=g) * K ** -0.5)
    bias = torch.randn(N, generator=g)
    x0 = torch.randn(M, N, generator=g) * 2.0
    x0[:, 7] += 60.0                                           # a massive-activation channel
    x0[:, :] += torch.randn(M, 1, generator=g)                 # and rows whose mean is not zero
    xs0 = hilo(x0)
    start = xs0[0].double() + xs0[1].double()                  # what the stream holds: x0 to 16 significand bits
    assert (start - x0.double()).abs().max().item() <= 2.0 ** -16 * 70
    ref = start + A.double() @ W.double().t() + bias.double()
    xs = xs0.clone().cuda()
    Ad, Wd, bd = A.to(torch.bfloat16).cuda(), W.to(torch.bfloat16).cuda(), bias.cuda()
    stats = _fold_resid(Ad, Wd, bd, xs, group32=g32)
    torch.cuda.synchronize()
    hi, lo = xs[0].cpu(), xs[1].cpu()
    xc = hi.double() + lo.double()
    assert torch.allclose(xc, ref, atol=2e-3, rtol=2e-5)      # fp32 sum, then the hi + lo split (2^-17 relative)
    assert (hi != ref.to(torch.bfloat16)).float().mean().item() < 2e-3      # hi IS the bf16 rounding of the new row (ties aside)
    assert (lo.float().abs() <= hi.float().abs() * 2.0 ** -8 + 1e-30).all()  # and lo what that rounding left
    rstd = stats[:M].cpu().double()
    want = 1.0 / torch.sqrt(ref.var(dim=1, unbiased=False) + 1e-5)
    assert ((rstd - want).abs() / want).max().item() <= 2e-5
    assert int(_counters(stats, M).abs().max()) == 0           # left at zero for the next launch
    # ... which is the same launch again on the same scratch: same bits
    xs2 = xs0.clone().cuda()
    stats2 = _fold_resid(Ad, Wd, bd, xs2, stats=stats.clone(), group32=g32)
    assert torch.equal(xs2, xs) and torch.equal(stats2[:M], stats[:M])


def test_fold_gemms_do_not_depend_on_the_row_count():
    """Rows 0..255 alone (128 x 128 tiles, two workgroups per CU) and as part of 12800 rows (160 x 256 tiles / the persistent
    kernel): the same bits out of both forms — statistics included."""
    g = torch.Generator().manual_seed(77)
    M, W_, F = 12800, 768, 3072
    A = bf16_round(torch.randn(M, W_, generator=g)).to(torch.bfloat16).cuda()
    Wq = bf16_round(torch.randn(3 * W_, W_, generator=g) * W_ ** -0.5).to(torch.bfloat16).cuda()
    bq = torch.randn(3 * W_, generator=g).cuda()
    rstd = (torch.rand(M, generator=g) * 0.5 + 0.1).cuda()
    for mode in (0, 1):
        big = _fold_bf16(A, Wq, bq, rstd, mode)
        for rows in (256, 2048, 6400):
            assert torch.equal(_fold_bf16(A[:rows].contiguous(), Wq, bq, rstd[:rows].contiguous(), mode), big[:rows]), (mode, rows)
    Ah = bf16_round(torch.randn(M, F, generator=g)).to(torch.bfloat16).cuda()
    Wp = bf16_round(torch.randn(W_, F, generator=g) * F ** -0.5).to(torch.bfloat16).cuda()
    bp = torch.randn(W_, generator=g).cuda()
    xs0 = hilo(torch.randn(M, W_, generator=g) * 2).cuda()
    xb = xs0.clone()
    sb = _fold_resid(Ah, Wp, bp, xb)
    for rows in (256, 2048, 6400):
        xs = xs0[:, :rows].contiguous()
        ss = _fold_resid(Ah[:rows].contiguous(), Wp, bp, xs)
        assert torch.equal(xs, xb[:, :rows]) and torch.equal(ss[:rows], sb[:rows]), rows


def test_fold_layernorm_weights_are_the_layernorm():
    """The packer's algebra on the CPU, in float64: rstd * (x W''^T) + b'' == Linear(LayerNorm(x))."""
    g = torch.Generator().manual_seed(3)
    x = torch.randn(9, 256, generator=g).double() * 3 + 1.5
    w, b = torch.randn(64, 256, generator=g), torch.randn(64, generator=g)
    gamma, beta = torch.rand(256, generator=g) + 0.5, torch.randn(256, generator=g)
    wf, bf = fold_layernorm(w, b, gamma, beta)
    ln = (x - x.mean(1, keepdim=True)) / torch.sqrt(x.var(1, unbiased=False, keepdim=True) + 1e-5) * gamma.double() + beta.double()
    want = ln @ w.double().t() + b.double()
    rstd = 1.0 / torch.sqrt(x.var(1, unbiased=False, keepdim=True) + 1e-5)
    got = rstd * (x @ wf.double().t()) + bf.double()
    assert (got - want).abs().max().item() <= 1e-4
    assert wf.double().sum(1).abs().max().item() <= 1e-4      # centred rows


@pytest.mark.parametrize("name,stress", [("vit_b32.npz", False), ("vit_b16.npz", False), ("vit_l14.npz", False),
                                         ("vit_b32_stress.npz", True), ("vit_l14_stress.npz", True)])
@pytest.mark.parametrize("fold", [2, 1, 0])
def test_vit_golden_in_both_modes(golden_dir, name, stress, fold):
    """The tower with and without the fold against the committed golden vectors (the oracle pinned to transformers' CLIP),
    seeded and checkpoint-like weights; the residual stream of the last block too.  The two modes are NOT bit-equal (the
    fold rounds gamma * W instead of the normalised row) — both sit within the same tolerance of the fp32 path."""
    spec, g, frames = load_golden(golden_dir, name)
    if fold == 2 and not (spec.tokens <= 64 and spec.heads == 12):
        pytest.skip("attention + out-projection as one kernel: up to 64 tokens, 12 heads")
    sd = (checkpoint_like_state_dict if stress else random_state_dict)(spec, int(g["weight_seed"]))
    eng = VitEngine(spec, sd, max_batch=frames.shape[0], ln_fold=fold)
    assert eng.spec.ln_fold == fold and eng.cfg.ln_fold == fold
    gold = torch.from_numpy(g["out"])
    out = eng.forward(vit_ref.normalize_u8(frames)).cpu()
    assert cosine(out, gold) >= 1 - COS_TOL, cosine(out, gold)
    assert 1 - cosine(out, gold) <= 3e-4                       # measured 1e-5 .. 1.2e-4; far inside the bar in both modes
    assert cosine(eng.forward(frames).cpu(), gold) >= 1 - COS_TOL
    taps = torch.from_numpy(g["taps"])
    x = eng.residual(frames.shape[0]).cpu().reshape(frames.shape[0], spec.tokens, spec.width)
    last = taps[-1] if taps.dim() == 3 else taps[-1][:, 0, :]
    assert cosine(x[:, 0, :], last) >= 1 - COS_TOL


def test_fold_mode_defaults_and_refusals():
    from wise_amd.feature.vit import spec_for
    from wise_amd.feature.siglip import SIGLIP_VISION

    b32 = spec_for("ViT-B-32", "openai")
    from wise_amd.feature.vit import DEFAULT_FOLD_B32
    assert VitEngine(b32, random_state_dict(b32, 0), max_batch=2).spec.ln_fold == DEFAULT_FOLD_B32 >= 1   # width 768: measured, on
    l14 = spec_for("ViT-L-14", "openai")
    assert VitEngine(l14, random_state_dict(l14, 0), max_batch=2).spec.ln_fold == 0
    with pytest.raises(ValueError):
        VitEngine(l14, random_state_dict(l14, 0), max_batch=2, ln_fold=2)                          # 257 tokens, 16 heads
    bad2 = _lib.VitConfig(224, 16, 768, 2, 12, 3072, 512, 0, 0, 2)                                 # 197 tokens
    import ctypes as C2
    assert _lib.lib().wise_vit_workspace_bytes(C2.byref(bad2), 1) == 0 and b"ln_fold = 2" in _lib.lib().wise_last_error()
    lib = _lib.lib()
    bad = _lib.VitConfig(224, 16, 768, 2, 12, 3072, 768, 1, 1, 1)                                 # the timm tower has no fold
    import ctypes as C
    assert lib.wise_vit_workspace_bytes(C.byref(bad), 1) == 0 and b"ln_fold" in lib.wise_last_error()


def _attn_oproj_pair(B, T, seed, fused):
    """One block's attention half on a hi + lo stream, as the two calls (attention, folded residual GEMM) or as the one kernel."""
    lib = _lib.lib()
    H, W = 12, 768
    M = B * T
    Mp = (M + 255) // 256 * 256
    g = torch.Generator().manual_seed(seed)
    qkv = torch.zeros(Mp, 3 * W, dtype=torch.bfloat16)
    qkv[:M] = (torch.randn(M, 3 * W, generator=g) * 1.5).to(torch.bfloat16)
    Wt = bf16_round(torch.randn(W, W, generator=g) * W ** -0.5).to(torch.bfloat16)
    bias = torch.randn(W, generator=g)
    x0 = torch.zeros(Mp, W)
    x0[:M] = torch.randn(M, W, generator=g) * 2 + 0.3
    xs = hilo(x0).cuda().contiguous()
    qkv, Wt, bias = qkv.cuda(), Wt.cuda(), bias.cuda()
    if fused:
        rstd = torch.full((Mp,), float("nan"), dtype=torch.float32, device="cuda")
        Wtiled = tile_out_proj(Wt)
        _lib.check(lib.wise_attention_oproj_fold(qkv.data_ptr(), B, T, H, Wtiled.data_ptr(), bias.data_ptr(), xs.data_ptr(), Mp * W,
                                                 rstd.data_ptr(), 1e-5, _lib.stream_ptr()), "wise_attention_oproj_fold")
    else:
        ao = torch.zeros(Mp, W, dtype=torch.bfloat16, device="cuda")
        _lib.check(lib.wise_attention_bf16(qkv.data_ptr(), B, T, H, ao.data_ptr(), _lib.stream_ptr()), "wise_attention_bf16")
        stats = _fold_resid(ao, Wt, bias, xs)
        rstd = stats[:Mp].clone()
    torch.cuda.synchronize()
    return xs.cpu(), rstd.cpu(), (qkv.cpu(), Wt.cpu(), bias.cpu(), x0)


@pytest.mark.parametrize("B,T", [(256, 50), (37, 50), (1, 50), (5, 64), (8, 33), (16, 17), (3, 1)])
def test_attention_and_out_projection_as_one_kernel(B, T):
    """wise_attention_oproj_fold against the two calls it replaces — BIT for bit (hi, lo and rstd of every real row; rows past
    B*T untouched) — and against a float64 reference of softmax(q k^T / 8) v W_o^T + b + x."""
    M = B * T
    xs1, r1, (qkv, Wt, bias, x0) = _attn_oproj_pair(B, T, 100 + B + T, fused=True)
    xs0, r0, _ = _attn_oproj_pair(B, T, 100 + B + T, fused=False)
    assert torch.equal(xs1[:, :M].view(torch.int16), xs0[:, :M].view(torch.int16))
    assert torch.equal(r1[:M].view(torch.int32), r0[:M].view(torch.int32))
    assert torch.equal(xs1[:, M:].view(torch.int16), hilo(x0)[:, M:].view(torch.int16))           # padding rows as they were
    assert torch.isnan(r1[M:]).all()
    q, k, v = (qkv[:M].double().reshape(B, T, 3, 12, 64).permute(2, 0, 3, 1, 4))
    o = torch.softmax(q @ k.transpose(-1, -2) / 8.0, dim=-1) @ v                                   # [B, 12, T, 64]
    o = o.permute(0, 2, 1, 3).reshape(M, 768)
    want = x0[:M].double() + o @ Wt.double().t() + bias.double()
    got = xs1[0, :M].double() + xs1[1, :M].double()
    assert (got - want).abs().max().item() <= 0.03 * max(1.0, want.abs().max().item() / 8)        # bf16 attention output
    want_rstd = 1.0 / torch.sqrt(got.var(1, unbiased=False) + 1e-5)
    assert ((r1[:M].double() - want_rstd).abs() / want_rstd).max().item() <= 1e-4


def test_one_kernel_form_does_not_depend_on_the_batch():
    """a frame's rows after wise_attention_oproj_fold are the same bits alone and inside a batch (one workgroup per frame)"""
    xs_all, r_all, (qkv, Wt, bias, x0) = _attn_oproj_pair(24, 50, 7, fused=True)
    lib = _lib.lib()
    for b in (0, 11, 23):
        q1 = torch.zeros(256, 2304, dtype=torch.bfloat16)
        q1[:50] = qkv[b * 50:(b + 1) * 50]
        x1 = torch.zeros(256, 768)
        x1[:50] = x0[b * 50:(b + 1) * 50]
        xs = hilo(x1).cuda().contiguous()
        rstd = torch.zeros(256, device="cuda")
        q1, Wd, bd = q1.cuda(), tile_out_proj(Wt).cuda(), bias.cuda()
        _lib.check(lib.wise_attention_oproj_fold(q1.data_ptr(), 1, 50, 12, Wd.data_ptr(), bd.data_ptr(), xs.data_ptr(), 256 * 768,
                                                 rstd.data_ptr(), 1e-5, _lib.stream_ptr()), "wise_attention_oproj_fold")
        torch.cuda.synchronize()
        assert torch.equal(xs.cpu()[:, :50].view(torch.int16), xs_all[:, b * 50:(b + 1) * 50].view(torch.int16))
        assert torch.equal(rstd.cpu()[:50], r_all[b * 50:(b + 1) * 50])


def test_tower_with_the_one_kernel_form_equals_the_two_kernel_fold():
    from wise_amd.feature.vit import spec_for
    spec = spec_for("ViT-B-32", "openai")
    sd = random_state_dict(spec, 3)
    x = torch.randn(37, 3, 224, 224, generator=torch.Generator().manual_seed(5)).cuda()
    a = VitEngine(spec, sd, max_batch=64, ln_fold=1).forward(x).cpu()
    b = VitEngine(spec, sd, max_batch=64, ln_fold=2).forward(x).cpu()
    assert torch.equal(a, b)
