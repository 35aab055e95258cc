#!/usr/bin/env python3
"""Derivation of the coefficients of `gelu_erf_fast` (csrc/wt_common.h): erf(|x| / sqrt 2) = 1 - 2^Q(|x|), Q a degree-8 polynomial
without constant term, fitted to log2(erfc) on [0, 3.92 sqrt 2] (beyond it erfc < 3e-8, and Q(t) t keeps falling monotonically, so no clamp) by iteratively
re-weighted least squares towards the minimax ABSOLUTE error of 1 - 2^Q, then checked in float32 Horner arithmetic against
scipy's erf on 4 M points.  CPU only:  python tools/fit_gelu.py"""
import numpy as np
from scipy.special import erf, erfc

T, DEG = 3.92, 8


def fit():
    t = 0.5 * T * (1 - np.cos(np.pi * (np.arange(4000) + 0.5) / 4000))
    y = np.log2(erfc(t))
    w, base = np.ones_like(t), erfc(t) * np.log(2)          # d(2^Q)/dQ = ln 2 * erfc: residuals in Q weighted into residuals in erf
    X = np.stack([t ** k for k in range(1, DEG + 1)], axis=1)
    for _ in range(60):
        c, *_ = np.linalg.lstsq(X * (w * base)[:, None], y * w * base, rcond=None)
        err = np.abs((1 - np.exp2(X @ c)) - erf(t))
        w = w * (1 + 4 * err / err.max())
        w /= w.mean()
    return c / np.sqrt(2.0) ** np.arange(1, DEG + 1)       # argument |x| instead of |x| / sqrt 2


def gelu_fast32(x, c):
    x = x.astype(np.float32)
    c = c.astype(np.float32)
    t = np.abs(x)
    q = np.full_like(t, c[-1])
    for k in range(DEG - 2, -1, -1):
        q = q * t + c[k]
    q = q * t
    e = np.exp2(q.astype(np.float64)).astype(np.float32)
    h = (np.float32(0.5) * x) * e
    return np.where(x >= 0, x - h, h)


if __name__ == "__main__":
    c = fit()
    x = np.linspace(-12, 12, 4000001)
    want = 0.5 * x * (1 + erf(x / np.sqrt(2.0)))
    got = gelu_fast32(x, c).astype(np.float64)
    err = np.abs(got - want)
    print("coefficients c1..c8 (float32):", ", ".join(f"{float(np.float32(v))!r}f" for v in c))
    t = np.concatenate([np.linspace(T * np.sqrt(2.0), 60, 100000), np.logspace(1.8, 19, 2000)]).astype(np.float32)
    q = np.full_like(t, np.float32(c[-1]))
    with np.errstate(over="ignore"):
        for k in range(DEG - 2, -1, -1):
            q = q * t + np.float32(c[k])
        qt = q * t
    assert not np.isnan(qt).any() and np.diff(qt[:100000]).max() < 0 and qt.max() < -24, "Q(t) t must keep falling beyond the fitted range"
    print(f"max |gelu_fast - gelu| on [-12, 12]: {err.max():.3e} at x = {x[err.argmax()]:.4f};  max relative to max(|x|, 1): {(err / np.maximum(np.abs(x), 1)).max():.3e}")
