"""The base encoder replaces `(x as f32) / C` by two FMAs around the rounded reciprocal (enc_div, rmj_encode.hip.h).  That is
only allowed if the result is the correctly rounded quotient for EVERY integer the encoder can feed it; this test proves it
with exact rational arithmetic for every (divisor, range) pair the kernel instantiates (and shows that the plain product with
the reciprocal would not do)."""
import re
from fractions import Fraction
from pathlib import Path

import numpy as np


def rn32(fr):
    """nearest binary32 (ties to even) of a Fraction, exactly"""
    if fr == 0:
        return Fraction(0)
    sign = -1 if fr < 0 else 1
    a = abs(fr)
    e = a.numerator.bit_length() - a.denominator.bit_length()
    if Fraction(2) ** e > a:
        e -= 1
    if Fraction(2) ** (e + 1) <= a:
        e += 1
    e = max(e, -126)
    ulp = Fraction(2) ** (e - 23)
    q = a / ulp
    n = q.numerator // q.denominator
    rem = q - n
    if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and (n & 1)):
        n += 1
    return sign * n * ulp


def instantiations():
    src = (Path(__file__).resolve().parent.parent / "riichienv_amd" / "csrc" / "rmj_encode.hip.h").read_text()
    pairs = sorted({(int(c), int(m)) for c, m in re.findall(r"enc_div<(\d+), (\d+)>\(", src)})
    assert len(pairs) >= 8, pairs
    return pairs


def test_two_fma_quotient_is_the_ieee_quotient_on_every_instantiated_range():
    naive_wrong = 0
    for c, xmax in instantiations():
        rc = rn32(Fraction(1, c))
        assert float(rc) == float(np.float32(1.0) / np.float32(c))
        for x in range(xmax + 1):
            xf = Fraction(x)
            q = rn32(xf * rc)                 # __fmul_rn(x, rc)
            r = rn32(xf - q * c)              # fmaf(-q, c, x): exact product and sum, one rounding
            got = rn32(q + r * rc)            # fmaf(r, rc, q)
            want = rn32(Fraction(x, c))
            assert float(want) == float(np.float32(x) / np.float32(c))     # rn32 agrees with IEEE division
            assert got == want, (c, x, float(got), float(want))
            naive_wrong += q != want
    assert naive_wrong > 1000     # the correction step is what makes it exact
