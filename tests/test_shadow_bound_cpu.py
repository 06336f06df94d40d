"""The error bound the two-stage search rests on (csrc/ip_topk.hip, shadow_eps), checked on the CPU: scoring
bf16-rounded rows against an fp32 query differs from the fp32 score by at most
    eps = |q| (max_r |x_r - bf16(x_r)|  +  d 2^-23 max_r |x_r|)
(Cauchy-Schwarz on the rounding residual, plus the f32 accumulation error of both dot products) — for unit rows, for
rows of wildly different norms, and for adversarial rows that sit just above powers of two (where round-to-nearest-even
loses the most: there the residual norm reaches its worst case 2^-8 |x|, and 2^-9 — half of it — would be unsound).
Then the threshold form's argument itself: every row of the exact top-k has approximate score >= s_A - 2 eps, where
s_A is the k-th best approximate score of ANY subset of the rows (the kernels use 128 evenly spaced chunks)."""
import numpy as np
import pytest
import torch


def bf16_round(x: np.ndarray) -> np.ndarray:
    return torch.from_numpy(x).to(torch.bfloat16).to(torch.float32).numpy()


@pytest.mark.parametrize("kind", ["unit", "mixed_norms", "worst_case_mantissas"])
@pytest.mark.parametrize("d", [64, 512, 1024])
def test_bf16_shadow_score_error_is_within_the_certificate_bound(kind, d):
    rng = np.random.default_rng(7 + d)
    n = 4096
    X = rng.standard_normal((n, d)).astype(np.float32)
    if kind == "unit":
        X /= np.linalg.norm(X, axis=1, keepdims=True)
    elif kind == "mixed_norms":
        X *= np.exp(rng.uniform(-6, 6, size=(n, 1))).astype(np.float32)
    else:
        # mantissa 1.0000000_1xxx...: the bf16 neighbour is almost half an ulp (2^-8 relative) away
        e = rng.integers(-8, 4, size=X.shape)
        X = (np.sign(X) * np.ldexp(1.0 + 2.0 ** -8 - 2.0 ** -20, e)).astype(np.float32)
    q = rng.standard_normal(d).astype(np.float32)
    Xb = bf16_round(X)
    approx = (Xb.astype(np.float32) * q).sum(axis=1, dtype=np.float32)         # f32 accumulation, like the kernels
    exact = (X * q).sum(axis=1, dtype=np.float32)
    truth = X.astype(np.float64) @ q.astype(np.float64)
    max_norm = float(np.linalg.norm(X.astype(np.float64), axis=1).max())
    max_err = float(np.linalg.norm(X.astype(np.float64) - Xb.astype(np.float64), axis=1).max())
    assert max_err <= 2.0 ** -8 * max_norm * (1 + 1e-9)
    if kind == "worst_case_mantissas":
        assert max_err > 2.0 ** -9 * max_norm          # the often-quoted 2^-9 is the AVERAGE case, not a bound
    else:
        assert max_err < 0.65 * 2.0 ** -8 * max_norm   # random mantissas: ~0.42 of the worst case on average
    eps = (max_err + d * 2.0 ** -23 * max_norm) * 1.0001 * float(np.linalg.norm(q.astype(np.float64)))
    assert np.abs(approx.astype(np.float64) - exact.astype(np.float64)).max() <= eps
    assert np.abs(approx.astype(np.float64) - truth).max() <= eps
    # the rounding term alone, row by row (Cauchy-Schwarz is what turns it into |q| |x|)
    per_row = np.abs((Xb.astype(np.float64) - X.astype(np.float64)) * q.astype(np.float64)).sum(axis=1)
    assert np.all(per_row <= 2.0 ** -8 * np.abs(X.astype(np.float64) * q.astype(np.float64)).sum(axis=1) * (1 + 1e-12))


@pytest.mark.parametrize("layout", ["iid", "runs_of_near_duplicates", "one_flat_run"])
def test_threshold_form_collects_every_row_of_the_exact_topk(layout):
    """wise_ip_topk_shadow_f32, one query: thr = (k-th best approximate score of the sample) - 2 eps; the rows with
    approximate score >= thr contain the exact top-k, whatever the data and wherever the sample was taken."""
    rng = np.random.default_rng(11)
    n, d, k = 40000, 128, 10
    X = rng.standard_normal((n, d)).astype(np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    q = rng.standard_normal(d).astype(np.float32)
    q /= np.linalg.norm(q)
    if layout == "runs_of_near_duplicates":
        base = X[::20].repeat(20, axis=0)[:n]
        X = base + 0.003 * rng.standard_normal((n, d)).astype(np.float32)
        X /= np.linalg.norm(X, axis=1, keepdims=True)
    elif layout == "one_flat_run":
        for r in range(17000, 17400):                     # 400 rows scoring 0.5 +- 1e-4: all within the bf16 error
            v = rng.standard_normal(d).astype(np.float32)
            v -= (v @ q) * q
            v /= np.linalg.norm(v)
            s0 = 0.5 + 1e-4 * rng.uniform(-1, 1)
            X[r] = s0 * q + np.sqrt(1 - s0 * s0) * v
    Xb = bf16_round(X)
    approx = (Xb * q).sum(axis=1, dtype=np.float32)
    exact = (X * q).sum(axis=1, dtype=np.float32)
    max_norm = float(np.linalg.norm(X.astype(np.float64), axis=1).max())
    max_err = float(np.linalg.norm(X.astype(np.float64) - Xb.astype(np.float64), axis=1).max())
    eps = (max_err + d * 2.0 ** -23 * max_norm) * 1.0001
    top = np.argsort(-exact, kind="stable")[:k]
    for sample in (np.arange(0, 2048), np.arange(n - 2048, n),                      # a corner of the index
                   np.concatenate([np.arange(c, c + 64) for c in range(0, n, n // 32)])):   # evenly spaced chunks
        s_a = np.sort(approx[sample])[-k]
        collected = np.where(approx >= s_a - 2 * eps)[0]
        assert set(top) <= set(collected)
        assert len(collected) >= k


@pytest.mark.parametrize("kind", ["unit", "mixed_norms", "worst_case_mantissas"])
@pytest.mark.parametrize("d", [512, 768])
def test_one_piece_query_bound_of_the_batched_scan(kind, d):
    """The batched shadow scan multiplies bf16 rows by ONE bf16 piece of the query (csrc/ip_topk_mfma.hip, LO = false);
    csrc/ip_topk.hip `query_eps(QMODE_ONE_PIECE)` then adds (max|x| + max residual) |q - bf16 q| to the row-rounding bound:
    |bf16(x) . bf16(q) - x . q| <= |bf16(x)| |q - bf16 q| + |x - bf16 x| |q|."""
    rng = np.random.default_rng(3 + d)
    n = 4096
    X = rng.standard_normal((n, d)).astype(np.float32)
    if kind == "unit":
        X /= np.linalg.norm(X, axis=1, keepdims=True)
    elif kind == "mixed_norms":
        X *= np.exp(rng.uniform(-6, 6, size=(n, 1))).astype(np.float32)
    else:
        e = rng.integers(-8, 4, size=X.shape)
        X = (np.sign(X) * np.ldexp(1.0 + 2.0 ** -8 - 2.0 ** -20, e)).astype(np.float32)
    for qkind in ("random", "worst_case_mantissas"):
        q = rng.standard_normal(d).astype(np.float32)
        if qkind == "worst_case_mantissas":
            q = (np.sign(q) * np.ldexp(1.0 + 2.0 ** -8 - 2.0 ** -20, rng.integers(-6, 2, size=d))).astype(np.float32)
        Xb, qb = bf16_round(X), bf16_round(q)
        approx = (Xb * qb).sum(axis=1, dtype=np.float32)                     # products exact in f32, f32 accumulation
        truth = X.astype(np.float64) @ q.astype(np.float64)
        n0 = float(np.linalg.norm(X.astype(np.float64), axis=1).max())
        n1 = float(np.linalg.norm(X.astype(np.float64) - Xb.astype(np.float64), axis=1).max())
        qn = float(np.linalg.norm(q.astype(np.float64)))
        qr = float(np.linalg.norm(q.astype(np.float64) - qb.astype(np.float64)))
        eps = (n1 + d * 2.0 ** -23 * n0) * 1.0001 * qn + (n0 + n1) * qr * 1.0001 + 1e-6 * qn * n0
        assert np.abs(approx.astype(np.float64) - truth).max() <= eps
        # and it is not vacuous: within a factor of a few of the two-piece bound on unit data
        if kind == "unit" and qkind == "random":
            assert eps < 4 * (n1 + d * 2.0 ** -23 * n0) * qn


def quantize_rows_i8(X: np.ndarray):
    """wise_ip_shadow_i8 restated: scale_r = max|x_r| / 127, c = clamp(rint(x * (127 / max)), +-127) (fp32 arithmetic)."""
    mx = np.abs(X).max(axis=1).astype(np.float32)
    scale = (mx / np.float32(127.0)).astype(np.float32)
    inv = np.where(mx > 0, np.float32(127.0) / np.where(mx > 0, mx, 1), 0).astype(np.float32)
    c = np.clip(np.rint(X * inv[:, None]), -127, 127).astype(np.int32)
    return c, scale


def quantize_query_i8(q: np.ndarray):
    """ShadowGroupI8::load_query restated: q^ = sq h + (sq / 254) l, two int8 pieces."""
    mq = np.float32(np.abs(q).max())
    sq = np.float32(mq / np.float32(127.0))
    sl = np.float32(sq / np.float32(254.0))
    inv = np.float32(127.0) / mq if mq > 0 else np.float32(0)
    invl = np.float32(254.0 * 127.0) / mq if mq > 0 else np.float32(0)
    h = np.clip(np.rint(q * inv), -127, 127).astype(np.float32)
    r = (q - sq * h).astype(np.float32)
    l = np.clip(np.rint(r * invl), -127, 127).astype(np.float32)
    return h.astype(np.int64), l.astype(np.int64), sq, sl


@pytest.mark.parametrize("kind", ["unit", "mixed_norms", "one_large_component", "peaky_query"])
@pytest.mark.parametrize("d", [64, 512, 1024])
def test_int8_shadow_score_error_is_within_the_bound_its_norms_carry(kind, d):
    """|q.x - s^| <= |q| (rho_max + sqrt(d) 1.6e-5 X^max) (+ the d 2^-23 X^max |q| slack shadow_eps adds): the residual of the
    rows by Cauchy-Schwarz, the residual of the two-piece int8 query by its worst case sqrt(d) sq / 508 with
    sq <= |q| / 127, integer sums exact."""
    rng = np.random.default_rng(70 + d)
    n = 4096
    X = rng.standard_normal((n, d)).astype(np.float32)
    if kind == "unit":
        X /= np.linalg.norm(X, axis=1, keepdims=True)
    elif kind == "mixed_norms":
        X *= np.exp(rng.uniform(-6, 6, size=(n, 1))).astype(np.float32)
    elif kind == "one_large_component":
        X[np.arange(n), rng.integers(0, d, n)] = 40.0      # the scale follows the outlier: everything else is coarse
    q = rng.standard_normal(d).astype(np.float32)
    if kind == "peaky_query":
        q[rng.integers(0, d, 3)] = 25.0
    c, scale = quantize_rows_i8(X)
    h, l, sq, sl = quantize_query_i8(q)
    assert np.abs(c).max() <= 127 and np.abs(h).max() <= 127 and np.abs(l).max() <= 127
    Xhat = scale[:, None].astype(np.float64) * c
    rho = np.linalg.norm(X.astype(np.float64) - Xhat, axis=1)
    assert np.all(rho <= scale.astype(np.float64) * np.sqrt(d) / 2 * (1 + 1e-5))          # half a step per component
    qhat = float(sq) * h + float(sl) * l
    qn = float(np.linalg.norm(q.astype(np.float64)))
    assert np.linalg.norm(q.astype(np.float64) - qhat) <= np.sqrt(d) * 1.6e-5 * qn
    H, L = c @ h, c @ l                                                                    # exact integers
    assert np.abs(H).max() < 2 ** 24 and np.abs(L).max() < 2 ** 24                         # exact as fp32, too
    approx = scale.astype(np.float64) * (float(sq) * H + float(sl) * L)
    truth = X.astype(np.float64) @ q.astype(np.float64)
    xmax, rmax = float(np.linalg.norm(Xhat, axis=1).max()), float(rho.max())
    norms1 = rmax + np.sqrt(d) * 1.6e-5 * xmax
    eps = (norms1 + d * 2.0 ** -23 * xmax) * 1.0001 * qn
    assert np.abs(approx - truth).max() <= eps
    if kind == "unit":
        assert rmax < 0.03                      # ~1-2 % of |x| for Gaussian rows (bf16: 0.2 %)
