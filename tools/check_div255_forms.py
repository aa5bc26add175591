"""Exhaustive proof behind the scalar tiers' pixel scaling (simd_dct_amd/csrc/mdct_kernels.hip: encode_block_pk, B4/B5): the
reference divides every pixel by 255 (simd_dct.cpp:222, :343: `px / 255.f`, an IEEE division, correctly rounded).  Division-free
forms, each evaluated in EXACT rational arithmetic with one round-to-nearest-even per IEEE operation, for all 256 byte values:
    mul            q = rn(x * c)                                  c  = rn(1/255)
    mul_fma        q = fma(x, c, rn(x * c2))                      c2 = rn(1/255 - c)      (two operations)
    newton         q0 = rn(x * c); r = fma(-255, q0, x); q = fma(r, c, q0)               (three operations)
CPU only, pure Python:   python3 tools/check_div255_forms.py > profiles/r05_div255_forms_exhaustive.log"""
from fractions import Fraction
import struct


def rn32(v):
    """round a Fraction to the nearest binary32 (ties to even); returns the Fraction of the float"""
    if v == 0:
        return Fraction(0)
    s = -1 if v < 0 else 1
    a = abs(v)
    e = a.numerator.bit_length() - a.denominator.bit_length()  # a ~ 2^e
    if Fraction(2) ** e > a:
        e -= 1
    e = max(e, -126)  # subnormals share the exponent of the smallest normal
    ulp = Fraction(2) ** (e - 23)
    n = a / ulp
    f = n.numerator // n.denominator
    rem = n - f
    if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and f % 2 == 1):
        f += 1
    return s * f * ulp


def bits(v):
    return struct.unpack("<I", struct.pack("<f", float(v)))[0]  # float(v) is exact: v is a binary32 value


C = rn32(Fraction(1, 255))
C2 = rn32(Fraction(1, 255) - C)


def forms(x):
    x = Fraction(x)
    q0 = rn32(x * C)
    yield "mul", q0
    yield "mul_fma", rn32(x * C + rn32(x * C2))
    r = rn32(x - 255 * q0)
    yield "newton", rn32(r * C + q0)


if __name__ == "__main__":
    bad = {}
    for x in range(256):
        want = rn32(Fraction(x, 255))
        for name, got in forms(x):
            bad.setdefault(name, [])
            if got != want:
                bad[name].append(x)
    print(f"c = rn(1/255) = {float(C)!r} (0x{bits(C):08x}), c2 = rn(1/255 - c) = {float(C2)!r} (0x{bits(C2):08x})")
    for name, b in bad.items():
        print(f"{name:8s} differs from rn(x / 255) for {len(b)} of 256 byte values {b[:8]}{' ...' if len(b) > 8 else ''}")
    assert not bad["newton"]
