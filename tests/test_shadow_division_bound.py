"""The bound behind the shadow march's skipped division (lol_kernel.h, soft_shadow): with a = 50 s and b = fl(res * t) in binary32,
   a > b   must imply   fl(a / t) >= res   for every 0 < res <= 1 and 0 <= t < 10^17 —
then v_min(res, a / t) is res and the wave need not divide.  Checked here on the CPU in exact rational arithmetic at the WORST a
(the float just above b), over normal, denormal and zero products, binade boundaries included.  (What the device computes is one
IEEE binary32 multiply and a compare; the GPU suite compares the frames that result with the oracle's.)"""
from fractions import Fraction

import numpy as np


def test_a_above_b_means_the_quotient_cannot_lower_the_minimum():
    rng = np.random.default_rng(20261005)
    n = 40000
    res = np.exp2(rng.uniform(-149, 0, n)).astype(np.float32)
    res[:200] = np.float32(1.0)
    res[200:400] = np.nextafter(np.float32(0), np.float32(1))                # the smallest denormal
    res[400:800] = np.exp2(rng.integers(-126, 0, 400)).astype(np.float32)     # powers of two: products on binade boundaries
    t = np.exp2(rng.uniform(-149, 56, n)).astype(np.float32)
    t[::97] = np.float32(0.0)
    t[1::97] = np.float32(1e17)
    t[400:800] = np.exp2(rng.integers(-100, 56, 400)).astype(np.float32)
    checked = 0
    with np.errstate(divide="ignore", over="ignore", under="ignore"):
        for r, tt in zip(res, t):
            assert 0 < r <= 1
            b = np.float32(r * tt)                                            # fl(res * t), as the device computes it
            a = np.nextafter(b, np.float32(np.inf))                           # the smallest a with a > b
            assert a > b and a > 0
            if tt == 0:
                assert np.isinf(np.float32(a) / np.float32(tt))               # +inf >= res
            else:
                assert Fraction(float(a)) > Fraction(float(r)) * Fraction(float(tt)), (r, tt, a)       # a > res * t exactly ...
                assert np.float32(a / tt) >= r, (r, tt, a)                    # ... hence the rounded quotient cannot be below res
            checked += 1
    assert checked == n


def test_the_test_is_tight():
    """a == b is NOT enough (so the strict '>' matters): where res * t was rounded DOWN, a = b lies below res * t."""
    r, tt = np.float32(0.3), np.float32(0.7)
    found = False
    for k in range(2000):
        b = np.float32(r * tt)
        if Fraction(float(b)) < Fraction(float(r)) * Fraction(float(tt)):
            found = True
            break
        tt = np.nextafter(tt, np.float32(1))
    assert found
