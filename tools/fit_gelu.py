"""Fit of act_gelu (csrc/gemm_bf16.hip): x * sigmoid(x (c0 + c1 x^2 + c2 x^4)) against 0.5 x (1 + erf(x / sqrt 2)),
minimax over [-8, 8] (x^2 clamped at 64 in the kernel).  Prints the coefficients as the kernel holds them (times -log2 e)."""
import numpy as np
from scipy.optimize import minimize
from scipy.special import erf

x = np.linspace(-8, 8, 40001)
g = 0.5 * x * (1 + erf(x / np.sqrt(2)))


def f(c, x):
    x2 = x * x
    return x / (1 + np.exp(-x * (c[0] + x2 * (c[1] + x2 * c[2]))))


r = minimize(lambda c: np.max(np.abs(f(c, x) - g)), [1.5957691, 0.0713548, 0.0], method="Nelder-Mead",
             options={"xatol": 1e-10, "fatol": 1e-12, "maxiter": 20000})
print("coefficients", r.x, "max |error|", r.fun)
print("kernel constants (x -log2 e):", [float(np.float32(-v * 1.4426950408889634)) for v in r.x])
print("tanh form for comparison:", np.max(np.abs(f([1.5957691, 0.0713548, 0.0], x) - g)))
