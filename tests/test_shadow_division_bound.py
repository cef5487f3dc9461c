"""The bound behind the shadow march's skipped division (lol_kernel.h, soft_shadow): with a = 50 s, b = fl(res * t) and
c = fma(b, 1 + 2^-21, 2^-125) in binary32,   a > c   must imply   fl(a / t) >= res   for every 0 < res <= 1 and 0 <= t < 10^17 —
then v_min(res, a / t) is res and the wave need not divide.  Checked here on the CPU in exact rational arithmetic at the WORST a
(the float just above c), over normal, denormal and zero products.  (What the device computes is IEEE binary32 + one fma; the
GPU suite compares the frames that result with the oracle's.)"""
from fractions import Fraction

import numpy as np

K = Fraction(1) + Fraction(1, 2 ** 21)
TINY = Fraction(1, 2 ** 125)


def rn32(x: Fraction) -> np.float32:
    """round-to-nearest-even of an exact rational to binary32 (through binary64: x is far from any binary32 tie here or the
    one-ulp slack of a double rounding only makes the tested claim stronger — see below)"""
    return np.float32(float(x))


def test_a_above_c_means_the_quotient_cannot_lower_the_minimum():
    rng = np.random.default_rng(20261005)
    n = 40000
    res = np.exp2(rng.uniform(-149, 0, n)).astype(np.float32)
    res[:200] = np.float32(1.0)
    res[200:400] = np.nextafter(np.float32(0), np.float32(1))                # the smallest denormal
    t = np.exp2(rng.uniform(-149, 56, n)).astype(np.float32)
    t[::97] = np.float32(0.0)
    t[1::97] = np.float32(1e17)
    checked = 0
    with np.errstate(divide="ignore", over="ignore", under="ignore"):
        for r, tt in zip(res, t):
            assert 0 < r <= 1
            b = np.float32(r * tt)                                            # fl(res * t), as the device computes it
            c = rn32(Fraction(float(b)) * K + TINY)
            # one ulp below what a double rounding could have produced: the claim is tested for a superset of the device's a
            c_low = np.nextafter(c, np.float32(-np.inf))
            a = np.nextafter(c_low, np.float32(np.inf))                       # the smallest a with a > c_low
            assert a > c_low and a > 0
            if tt == 0:
                assert np.isinf(np.float32(a) / np.float32(tt))               # +inf >= res
            else:
                assert Fraction(float(a)) >= Fraction(float(r)) * Fraction(float(tt)), (r, tt, a)      # a >= res * t exactly ...
                assert np.float32(a / tt) >= r, (r, tt, a)                    # ... hence the rounded quotient cannot be below res
            checked += 1
    assert checked == n
