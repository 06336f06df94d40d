"""-m gpu: wise_mlp_stream — x += fc2(GELU(fc1(h))) of a Swin block of MS-CLAP's HTSAT (stages 2 and 3) as one kernel whose
hidden activations never leave the register file (msclap HTSAT SwinTransformerBlock.mlp as reached from
src/feature/microsoft_clap.py:49-50) — through the C ABI against a float64 torch reference and against the two GEMM calls it
replaces."""
import pytest
import torch

from tests.test_gpu_vit import bf16_round
from wise_amd import _lib
from wise_amd.feature.htsat import mlp_stream_weights

pytestmark = pytest.mark.gpu


def _inputs(M, C, seed):
    g = torch.Generator().manual_seed(seed)
    h = bf16_round(torch.randn(M, C, generator=g))
    w1 = bf16_round(torch.randn(4 * C, C, generator=g) * C ** -0.5)
    w2 = bf16_round(torch.randn(C, 4 * C, generator=g) * (4 * C) ** -0.5)
    b1, b2 = torch.randn(4 * C, generator=g) * 0.3, torch.randn(C, generator=g) * 0.3
    x = torch.randn(M, C, generator=g) * 2
    return h, w1, w2, b1, b2, x


def _stream(h, w1, w2, b1, b2, x):
    lib = _lib.lib()
    M, C = h.shape
    hd, ws = h.to(torch.bfloat16).cuda(), mlp_stream_weights(w1, w2).to(torch.bfloat16).cuda()
    b1d, b2d, xd = b1.cuda(), b2.cuda(), x.clone().cuda()
    _lib.check(lib.wise_mlp_stream(hd.data_ptr(), ws.data_ptr(), b1d.data_ptr(), b2d.data_ptr(), xd.data_ptr(), M, C,
                                   _lib.stream_ptr()), "wise_mlp_stream")
    torch.cuda.synchronize()
    return xd.cpu()


def _two_gemms(h, w1, w2, b1, b2, x):
    lib = _lib.lib()
    M, C = h.shape
    hd, w1d, w2d = h.to(torch.bfloat16).cuda(), w1.to(torch.bfloat16).cuda(), w2.to(torch.bfloat16).cuda()
    b1d, b2d, xd = b1.cuda(), b2.cuda(), x.clone().cuda()
    a = torch.empty(M, 4 * C, dtype=torch.bfloat16, device="cuda")
    st = _lib.stream_ptr()
    _lib.check(lib.wise_gemm_bf16(hd.data_ptr(), w1d.data_ptr(), b1d.data_ptr(), M, 4 * C, C, 2, a.data_ptr(), st), "fc1")
    _lib.check(lib.wise_gemm_bf16(a.data_ptr(), w2d.data_ptr(), b2d.data_ptr(), M, C, 4 * C, 3, xd.data_ptr(), st), "fc2")
    torch.cuda.synchronize()
    return xd.cpu()


@pytest.mark.parametrize("M,C", [(128, 384), (256, 384), (32768, 384), (128, 192), (1024, 192), (131072, 192)])
def test_mlp_stream_against_float64_and_the_two_gemms(M, C):
    h, w1, w2, b1, b2, x = _inputs(M, C, M + C)
    got = _stream(h, w1, w2, b1, b2, x)
    n = min(M, 2048)                                               # float64 reference on the first rows, all rows against the GEMMs
    hid = h[:n].double() @ w1.double().t() + b1.double()
    hid = bf16_round((0.5 * hid * (1 + torch.erf(hid / 2 ** 0.5))).float()).double()
    want = x[:n].double() + hid @ w2.double().t() + b2.double()
    # the hidden values are rounded to bf16 in both forms (one rounding may differ by an ulp where GELU's 2.6e-5 fit error
    # crosses a rounding boundary): a few 1e-3 on sums of 4C terms of size ~1/sqrt(4C)
    assert (got[:n].double() - want).abs().max().item() <= 2e-2
    assert (got[:n].double() - want).abs().mean().item() <= 1.5e-3
    ref = _two_gemms(h, w1, w2, b1, b2, x)
    assert (got - ref).abs().max().item() <= 2e-2
    assert (got - ref).abs().mean().item() <= 1e-3


def test_mlp_stream_rows_do_not_depend_on_the_batch():
    h, w1, w2, b1, b2, x = _inputs(1024, 384, 5)
    whole = _stream(h, w1, w2, b1, b2, x)
    part = _stream(h[256:384], w1, w2, b1, b2, x[256:384])
    assert torch.equal(whole[256:384], part)


def test_mlp_stream_refuses_other_shapes():
    lib = _lib.lib()
    t = torch.zeros(64, device="cuda")
    assert lib.wise_mlp_stream(t.data_ptr(), t.data_ptr(), t.data_ptr(), t.data_ptr(), t.data_ptr(), 128, 768, _lib.stream_ptr()) != 0
    assert b"mlp_stream" in lib.wise_last_error()
    assert lib.wise_mlp_stream(t.data_ptr(), t.data_ptr(), t.data_ptr(), t.data_ptr(), t.data_ptr(), 64, 192, _lib.stream_ptr()) != 0


@pytest.mark.parametrize("M,C", [(256, 384), (32768, 384), (128, 192), (131072, 192)])
def test_mlp_stream_with_the_layernorm_inside(M, C):
    """wise_mlp_stream_ln: h = LayerNorm(x) computed in the kernel — against wise_layernorm_f32_bf16 + wise_mlp_stream (the rows may
    differ where a normalised value sits on a bf16 rounding boundary: the statistics are summed in another order)"""
    lib = _lib.lib()
    _, w1, w2, b1, b2, x = _inputs(M, C, M + C + 1)
    g = torch.Generator().manual_seed(3)
    lnw, lnb = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2
    x = x * (torch.rand(M, 1, generator=g) * 3 + 0.2) + torch.randn(M, 1, generator=g)          # rows of different scale and offset
    ws = mlp_stream_weights(w1, w2).to(torch.bfloat16).cuda()
    b1d, b2d, lwd, lbd = b1.cuda(), b2.cuda(), lnw.cuda(), lnb.cuda()
    st = _lib.stream_ptr()
    xa, xb = x.clone().cuda(), x.clone().cuda()
    h = torch.empty(M, C, dtype=torch.bfloat16, device="cuda")
    _lib.check(lib.wise_layernorm_f32_bf16(xa.data_ptr(), lwd.data_ptr(), lbd.data_ptr(), M, C, 1e-5, h.data_ptr(), st), "ln")
    _lib.check(lib.wise_mlp_stream(h.data_ptr(), ws.data_ptr(), b1d.data_ptr(), b2d.data_ptr(), xa.data_ptr(), M, C, st), "mlp")
    _lib.check(lib.wise_mlp_stream_ln(lwd.data_ptr(), lbd.data_ptr(), 1e-5, ws.data_ptr(), b1d.data_ptr(), b2d.data_ptr(),
                                      xb.data_ptr(), M, C, st), "mlp_ln")
    torch.cuda.synchronize()
    d = (xa - xb).abs().cpu()
    assert d.max().item() <= 3e-2 and d.mean().item() <= 2e-4
    n = min(M, 1024)
    xd = x[:n].double()
    hn = (xd - xd.mean(1, keepdim=True)) / torch.sqrt(xd.var(1, unbiased=False, keepdim=True) + 1e-5) * lnw.double() + lnb.double()
    hid = bf16_round(hn.float()).double() @ w1.double().t() + b1.double()
    hid = bf16_round((0.5 * hid * (1 + torch.erf(hid / 2 ** 0.5))).float()).double()
    want = xd + hid @ w2.double().t() + b2.double()
    assert (xb.cpu()[:n].double() - want).abs().max().item() <= 4e-2
    assert (xb.cpu()[:n].double() - want).abs().mean().item() <= 2e-3
