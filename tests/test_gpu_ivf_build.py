"""-m gpu: the IVF build kernels (csrc/ivf_build.hip: faiss Clustering / IndexIVF.add as reached from
src/index/feature_search_index.py:53-76) through the C ABI against torch / numpy restatements: nearest-centroid argmax with ties,
the stable grouping by list (radix sort), per-list sums in a fixed order, normalisation, gathers."""
import numpy as np
import pytest
import torch

from wise_amd import _lib

pytestmark = pytest.mark.gpu


def test_argmax_rows_with_ties_and_nans():
    lib = _lib.lib()
    g = torch.Generator().manual_seed(1)
    s = torch.randn(777, 3163, generator=g)
    s[5, 100] = s[5, 2000] = 9.0                       # a tie: the lower column
    s[6] = float("nan")
    s[7, :] = -float("inf")
    sd = s.cuda()
    out = torch.empty(777, dtype=torch.int64, device="cuda")
    _lib.check(lib.wise_ivf_argmax(sd.data_ptr(), 777, 3163, out.data_ptr(), _lib.stream_ptr()), "argmax")
    got = out.cpu()
    want = s.argmax(dim=1)
    keep = torch.ones(777, dtype=torch.bool)
    keep[6] = False
    assert torch.equal(got[keep], want[keep]) and got[5] == 100 and got[6] == 0 and got[7] == 0


@pytest.mark.parametrize("n,nlist", [(0, 7), (1, 1), (1000, 7), (5000, 300), (300000, 5480), (2_000_000, 70000)])
def test_group_by_list_is_a_stable_sort(n, nlist):
    lib = _lib.lib()
    rng = np.random.default_rng(n + nlist)
    a = rng.integers(0, nlist, n).astype(np.int64)
    if n > 10:
        a[: n // 3] = a[0]                              # a heavy list
    ad = torch.from_numpy(a).cuda()
    order = torch.empty(n, dtype=torch.int64, device="cuda")
    off = torch.empty(nlist + 1, dtype=torch.int64, device="cuda")
    cnt = torch.empty(nlist, dtype=torch.int64, device="cuda")
    ws = torch.empty(lib.wise_ivf_group_workspace_bytes(n, nlist), dtype=torch.uint8, device="cuda")
    _lib.check(lib.wise_ivf_group(ad.data_ptr(), n, nlist, order.data_ptr(), off.data_ptr(), cnt.data_ptr(), ws.data_ptr(), ws.numel(),
                                  _lib.stream_ptr()), "group")
    torch.cuda.synchronize()
    want = np.argsort(a, kind="stable")
    counts = np.bincount(a, minlength=nlist)
    assert np.array_equal(order.cpu().numpy(), want)
    assert np.array_equal(cnt.cpu().numpy(), counts)
    assert np.array_equal(off.cpu().numpy(), np.concatenate([[0], np.cumsum(counts)]))


def test_list_sums_normalise_gather_expand_reseed():
    lib = _lib.lib()
    st = _lib.stream_ptr()
    n, d, nlist = 20000, 512, 97
    g = torch.Generator().manual_seed(3)
    x = torch.randn(n, d, generator=g)
    a = torch.randint(0, nlist - 2, (n,), generator=g)                 # the last two lists stay empty
    order = torch.argsort(a, stable=True)
    counts = torch.bincount(a, minlength=nlist)
    off = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(counts, 0)])
    xd, od, fd = x.cuda(), order.cuda(), off.cuda()
    sums = torch.full((nlist, d), float("nan"), device="cuda")
    _lib.check(lib.wise_ivf_list_sums(xd.data_ptr(), od.data_ptr(), fd.data_ptr(), nlist, d, sums.data_ptr(), st), "sums")
    want = torch.zeros(nlist, d, dtype=torch.float64).index_add_(0, a, x.double())
    assert (sums.cpu().double() - want).abs().max().item() <= 1e-3
    again = torch.empty_like(sums)
    _lib.check(lib.wise_ivf_list_sums(xd.data_ptr(), od.data_ptr(), fd.data_ptr(), nlist, d, again.data_ptr(), st), "sums")
    assert torch.equal(sums, again)                                     # a fixed order of addition: the same bits every time
    e = torch.tensor([nlist - 2, nlist - 1], device="cuda")
    dn = torch.tensor([3, 4], device="cuda")
    _lib.check(lib.wise_ivf_reseed(sums.data_ptr(), e.data_ptr(), dn.data_ptr(), 2, d, st), "reseed")
    sc = sums.cpu()
    assert torch.allclose(sc[nlist - 2], sc[3] * (1 + 1e-3 * torch.sign(sc[3])))
    c = torch.empty_like(sums)
    _lib.check(lib.wise_ivf_normalize_rows(sums.data_ptr(), nlist, d, c.data_ptr(), st), "normalize")
    assert torch.allclose(c.cpu(), sc / sc.norm(dim=1, keepdim=True).clamp_min(1e-20), atol=1e-6)
    out = torch.empty(n, d, device="cuda")
    _lib.check(lib.wise_ivf_gather_rows(xd.data_ptr(), od.data_ptr(), n, d, out.data_ptr(), st), "gather")
    assert torch.equal(out.cpu(), x[order])
    ids = torch.arange(n, dtype=torch.int64) * 7 + 1
    oi = torch.empty(n, dtype=torch.int64, device="cuda")
    idd = ids.cuda()
    _lib.check(lib.wise_ivf_gather_i64(idd.data_ptr(), od.data_ptr(), n, oi.data_ptr(), st), "gather_i64")
    assert torch.equal(oi.cpu(), ids[order])
    ex = torch.empty(n, dtype=torch.int64, device="cuda")
    _lib.check(lib.wise_ivf_expand_lists(fd.data_ptr(), nlist, ex.data_ptr(), st), "expand")
    assert torch.equal(ex.cpu(), torch.repeat_interleave(torch.arange(nlist), counts))
