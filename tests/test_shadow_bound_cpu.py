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
