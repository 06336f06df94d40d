"""-m gpu: wise_swin_qkv_attn — norm1 + QKV projection + window attention of a Swin block of MS-CLAP's HTSAT (stages 2 / 3) as one
kernel (msclap HTSAT WindowAttention as reached from src/feature/microsoft_clap.py:49-50) — through the C ABI against a float64
torch restatement of the shifted-window attention (roll, window partition, relative-position bias, region mask)."""
import pytest
import torch

from tests.test_gpu_vit import bf16_round
from wise_amd import _lib
from wise_amd.feature.htsat import swin_qkv_stream

pytestmark = pytest.mark.gpu


def _reference(x, lnw, lnb, w, bq, relb, B, H, C, shift):
    heads = C // 24
    xd = x.double().reshape(B, H, H, C)
    hn = (xd - xd.mean(-1, keepdim=True)) / torch.sqrt(xd.var(-1, unbiased=False, keepdim=True) + 1e-5) * lnw.double() + lnb.double()
    hn = bf16_round(hn.float()).double()
    qkv = bf16_round((hn @ w.double().t() + bq.double()).float()).double()             # [B,H,H,3C] (the GEMM's bf16 output)
    if shift:
        qkv = torch.roll(qkv, shifts=(-shift, -shift), dims=(1, 2))
    nw = H // 8
    win = qkv.reshape(B, nw, 8, nw, 8, 3, heads, 24).permute(0, 1, 3, 5, 6, 2, 4, 7).reshape(B * nw * nw, 3, heads, 64, 24)
    q, k, v = win[:, 0], win[:, 1], win[:, 2]
    att = q @ k.transpose(-1, -2) * 24 ** -0.5 + relb.double()[None]
    if shift:
        img = torch.zeros(H, H)
        cnt = 0
        for hs in (slice(0, -8), slice(-8, -shift), slice(-shift, None)):
            for wsl in (slice(0, -8), slice(-8, -shift), slice(-shift, None)):
                img[hs, wsl] = cnt
                cnt += 1
        mw = img.reshape(nw, 8, nw, 8).permute(0, 2, 1, 3).reshape(nw * nw, 64)
        mask = (mw[:, None, :] != mw[:, :, None]).double() * -100.0                         # [nwin, 64, 64]
        att = att.reshape(B, nw * nw, heads, 64, 64) + mask[None, :, None]
        att = att.reshape(B * nw * nw, heads, 64, 64)
    o = torch.softmax(att, dim=-1) @ v                                                      # [Bw, heads, 64, 24]
    o = o.reshape(B, nw, nw, heads, 8, 8, 24).permute(0, 1, 4, 2, 5, 3, 6).reshape(B, H, H, C)
    if shift:
        o = torch.roll(o, shifts=(shift, shift), dims=(1, 2))
    return o.reshape(B * H * H, C)


@pytest.mark.parametrize("B,H,C,shift", [(2, 16, 384, 0), (2, 16, 384, 4), (1, 32, 192, 0), (2, 32, 192, 4), (128, 16, 384, 4), (64, 32, 192, 4), (3, 16, 192, 4), (2, 8, 192, 0), (6, 8, 192, 4)])
def test_swin_qkv_attn_against_float64(B, H, C, shift):
    lib = _lib.lib()
    heads = C // 24
    g = torch.Generator().manual_seed(B + H + C + shift)
    M = B * H * H
    x = torch.randn(M, C, generator=g) * (torch.rand(M, 1, generator=g) * 2 + 0.3) + torch.randn(M, 1, generator=g) * 0.5
    lnw, lnb = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2
    w = bf16_round(torch.randn(3 * C, C, generator=g) * C ** -0.5)
    bq = torch.randn(3 * C, generator=g) * 0.3
    relb = torch.randn(heads, 64, 64, generator=g) * 0.5
    ws, bs = swin_qkv_stream(w, bq)
    xd, ws, bs = x.cuda(), ws.to(torch.bfloat16).cuda(), bs.cuda()
    o = torch.full((M, C), float("nan"), dtype=torch.bfloat16, device="cuda")
    lw, lb, rb = lnw.cuda(), lnb.cuda(), relb.contiguous().cuda()
    _lib.check(lib.wise_swin_qkv_attn(xd.data_ptr(), lw.data_ptr(), lb.data_ptr(), 1e-5, ws.data_ptr(), bs.data_ptr(), rb.data_ptr(),
                                      o.data_ptr(), B, H, C, shift, _lib.stream_ptr()), "wise_swin_qkv_attn")
    torch.cuda.synchronize()
    n = min(B, 4)
    want = _reference(x[: n * H * H], lnw, lnb, w, bq, relb, n, H, C, shift)
    got = o.cpu()[: n * H * H].double()
    assert torch.isfinite(o.float()).all()
    err = (got - want).abs()
    assert err.max().item() <= 0.06 and err.mean().item() <= 4e-3, (err.max().item(), err.mean().item())   # bf16 q / k / v / p / output


def test_swin_qkv_attn_refuses_other_shapes():
    lib = _lib.lib()
    t = torch.zeros(64, device="cuda")
    p = t.data_ptr()
    assert lib.wise_swin_qkv_attn(p, p, p, 1e-5, p, p, p, p, 2, 16, 768, 0, _lib.stream_ptr()) != 0
    assert b"swin_qkv_attn" in lib.wise_last_error()
    assert lib.wise_swin_qkv_attn(p, p, p, 1e-5, p, p, p, p, 1, 8, 384, 0, _lib.stream_ptr()) != 0      # one window: odd count
