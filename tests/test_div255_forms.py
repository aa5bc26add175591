"""The scalar tiers' pixel scaling issues  fma(x, c, rn(x * c2))  where the reference writes  px / 255.f  (simd_dct.cpp:222, :343).
Exhaustive over all 256 byte values, in exact rational arithmetic with one round-to-nearest-even per IEEE operation
(tools/check_div255_forms.py; log: profiles/r05_div255_forms_exhaustive.log), cross-checked against numpy's float32 division, and
the two constants the host passes to the kernel (mdct_api.hip) are the ones the proof is about."""
import os
import sys
from fractions import Fraction

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_div255_forms as D  # noqa: E402


def test_two_operation_form_equals_the_division_for_every_byte():
    wrong_plain = 0
    for x in range(256):
        want = D.rn32(Fraction(x, 255))
        got = dict(D.forms(x))
        assert got["mul_fma"] == want, x  # what the kernel issues: v_pk_mul_f32 + v_pk_fma_f32
        assert got["newton"] == want, x   # the three-operation form: also exact, one operation more
        wrong_plain += got["mul"] != want
        # the checker's rounding is IEEE's: numpy's float32 division agrees with rn32(x / 255)
        assert float(np.float32(x) / np.float32(255)) == float(want), x
    assert wrong_plain == 126  # why a plain multiply by 1/255 will not do


def test_the_constants_the_host_passes_are_the_proven_ones():
    c = np.float32(1) / np.float32(255)                  # mdct_api.hip: 1.f / 255.f
    c2 = np.float32(1.0 / 255.0 - float(c))              # mdct_api.hip: (float)(1.0 / 255.0 - (double)(1.f / 255.f))
    assert Fraction(float(c)) == D.C and Fraction(float(c2)) == D.C2
    assert D.bits(D.C) == 0x3B808081 and D.bits(D.C2) == 0xAF7EFEFF
    src = open(os.path.join(ROOT, "simd_dct_amd", "csrc", "mdct_api.hip")).read()
    assert "{1.f / 255.f, (float)(1.0 / 255.0 - (double)(1.f / 255.f))}" in src
    kern = open(os.path.join(ROOT, "simd_dct_amd", "csrc", "mdct_kernels.hip")).read()
    assert "div_tab" not in kern and "px_div255" not in kern  # the LDS quotient table of rounds 2-4 is gone
