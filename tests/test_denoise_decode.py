"""K3 decodes UNORM8 / SNORM8 guides with q0 = c*r, q = fma(fma(-D, q0, c), r, q0), r = RN(1/D), instead of the
IEEE division c/D the spec (and the oracle) use.  The two agree for every one of the 256 codes; this is the
exhaustive check, in exact rational arithmetic (csrc/vrt_device.hip: decode_unorm8 / decode_snorm8)."""
import math
from fractions import Fraction as F

import numpy as np


def rn32(x: F) -> np.float32:
    if x == 0:
        return np.float32(0.0)
    s, a = (-1 if x < 0 else 1), abs(x)
    e = math.floor(math.log2(float(a)))
    while F(2) ** e > a:
        e -= 1
    while F(2) ** (e + 1) <= a:
        e += 1
    ulp = F(2) ** (e - 23)
    q = a / ulp
    n = q.numerator // q.denominator
    rem = q - n
    if rem > F(1, 2) or (rem == F(1, 2) and n % 2 == 1):
        n += 1
    return np.float32(float(s * n * ulp))


def fma32(a, b, c):
    return rn32(F(float(a)) * F(float(b)) + F(float(c)))


def test_fma_refined_reciprocal_equals_ieee_division_for_all_codes():
    for D, codes in ((255, range(0, 256)), (127, range(-128, 128))):
        r = np.float32(1.0) / np.float32(D)
        for c in codes:
            cf = np.float32(c)
            exact = rn32(F(c, D))
            assert exact == np.float32(cf / np.float32(D))
            q0 = np.float32(cf * r)
            q = fma32(fma32(np.float32(-D), q0, cf), r, q0)
            assert q == exact, (D, c)
