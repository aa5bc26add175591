"""Exhaustive proof behind the scalar tiers' quantiser (simd_dct_amd/csrc/mdct_kernels.hip: encode_block_pk, B4/B5):
for EVERY float32 c in [0, 1] (2^30 + 1 values), the reference's  (uint8_t)roundf(c * 255.f)   (simd_dct.cpp:245, :362; c is the
clamped value) equals  trunc(rn(c * 255 + h))  for h = pred(0.5), fused (one rounding: what the kernel issues as v_pk_fma_f32 +
v_cvt_u32_f32) and unfused; with h = 0.5 both forms fail for exactly one c.   CPU only, numpy, ~2 minutes:
    python3 tools/check_roundf_forms.py > profiles/r04_roundf_forms_exhaustive.log"""
import numpy as np

N = 0x3F800001  # bit patterns of the floats 0.0 .. 1.0
STEP = 1 << 26
h = np.nextafter(np.float32(0.5), np.float32(0))
forms = {"fma(c,255,0.5)": 0, "rn(c*255)+0.5": 0, "fma(c,255,pred(0.5))": 0, "rn(c*255)+pred(0.5)": 0}
first = {k: [] for k in forms}
for a in range(0, N, STEP):
    v = np.arange(a, min(N, a + STEP), dtype=np.uint32).view(np.float32)
    v64 = v.astype(np.float64)
    x = v * np.float32(255.0)  # the reference's float product
    x64 = x.astype(np.float64)
    fl = np.floor(x64)
    want = fl + ((x64 - fl) >= 0.5)  # roundf for x >= 0: half away from zero
    got = {
        "fma(c,255,0.5)": (v64 * 255.0 + 0.5).astype(np.float32),  # c * 255 is exact in float64 (24 + 8 bits): one rounding
        "rn(c*255)+0.5": (x + np.float32(0.5)).astype(np.float32),
        "fma(c,255,pred(0.5))": (v64 * 255.0 + float(h)).astype(np.float32),
        "rn(c*255)+pred(0.5)": (x + h).astype(np.float32),
    }
    for k, s in got.items():
        bad = np.nonzero(np.trunc(s.astype(np.float64)) != want)[0]
        forms[k] += len(bad)
        if len(bad) and len(first[k]) < 3:
            first[k] += [(hex(int(v[i:i + 1].view(np.uint32)[0])), float(x[i])) for i in bad[:3]]
print(f"floats checked: {N} (0.0 .. 1.0 inclusive), pred(0.5) = {float(h)!r}")
for k in forms:
    print(f"{k:24s} mismatches vs roundf(rn(c*255)): {forms[k]}  {first[k]}")
assert forms["fma(c,255,pred(0.5))"] == 0 and forms["rn(c*255)+pred(0.5)"] == 0
