"""The error bound the two-stage search's certificate rests on (csrc/ip_topk.hip, rescore_certify_kernel), checked on the
CPU: scoring bf16-rounded rows against an fp32 query differs from the fp32 score by at most
(2^-8 + d 2^-23) |q| max|x| — for unit rows, for rows of wildly different norms, and for adversarial rows that sit just
above powers of two (where round-to-nearest-even loses the most)."""
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
    eps = (2.0 ** -8 + d * 2.0 ** -23) * 1.0001 * float(np.linalg.norm(q.astype(np.float64))) * max_norm
    assert np.abs(approx.astype(np.float64) - exact.astype(np.float64)).max() <= eps
    assert np.abs(approx.astype(np.float64) - truth).max() <= eps
    # the rounding term alone, row by row (Cauchy-Schwarz is what turns it into |q| |x|)
    per_row = np.abs((Xb.astype(np.float64) - X.astype(np.float64)) * q.astype(np.float64)).sum(axis=1)
    assert np.all(per_row <= 2.0 ** -8 * np.abs(X.astype(np.float64) * q.astype(np.float64)).sum(axis=1) * (1 + 1e-12))
