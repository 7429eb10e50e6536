#!/usr/bin/env python3
"""Coefficients of the sky-texel fast path's atan polynomial (csrc/vrt_sky.h): atan(t) = t * P(t^2) on [0, 1], P of degree 8,
fitted by Lawson-weighted least squares on Chebyshev nodes in float64, rounded to float32, and checked in float32 Horner
arithmetic (fused multiply-adds, as the kernel evaluates it) on a dense sample of [0, 1].  Prints the table and the errors."""
import numpy as np


def lawson(f, deg, lo, hi, n=6000, iters=300):
    x = np.cos(np.pi * (np.arange(n) + 0.5) / n) * (hi - lo) / 2 + (hi + lo) / 2
    y = f(x)
    w = np.ones(n)
    V = np.vander(x, deg + 1, increasing=True)
    for _ in range(iters):
        c = np.linalg.lstsq(V * w[:, None] ** 0.5, y * w ** 0.5, rcond=None)[0]
        e = np.abs(V @ c - y)
        w = w * (e / e.max() + 1e-3)
        w /= w.sum()
    return c, e.max()


def atan_over_t(z):
    s = np.sqrt(np.maximum(z, 1e-300))
    return np.where(z > 1e-30, np.arctan(s) / s, 1.0)


def horner32(c32, t):
    """t * P(t^2) in float32 with fused multiply-adds (emulated in float64 and rounded once per operation)."""
    t = t.astype(np.float32)
    z = (t.astype(np.float64) * t.astype(np.float64)).astype(np.float32)
    p = np.full_like(t, c32[-1])
    for c in c32[-2::-1]:
        p = (p.astype(np.float64) * z.astype(np.float64) + np.float64(c)).astype(np.float32)
    return (p.astype(np.float64) * t.astype(np.float64)).astype(np.float32)


if __name__ == "__main__":
    for deg in (7, 8):
        c, e = lawson(atan_over_t, deg, 0.0, 1.0)
        c32 = c.astype(np.float32)
        t = np.concatenate([np.linspace(0, 1, 4_000_001), np.random.default_rng(1).random(4_000_000)]).astype(np.float32)
        got = horner32(c32, t).astype(np.float64)
        err = np.abs(got - np.arctan(t.astype(np.float64)))
        print(f"degree {deg}: fit error {e:.3e} (relative to t); float32 evaluation: max |error| {err.max():.3e} rad at t = {t[err.argmax()]:.6f}")
        print("  coefficients (float32, lowest first):")
        print("   ", ", ".join(f"{float(x):.9e}f" for x in c32))
        print("   ", ", ".join(hex(int(x.view(np.uint32))) for x in c32))
