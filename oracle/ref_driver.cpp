// ref_driver.cpp -- extern "C" handles onto the REAL reference, for validating the oracle.
//
// TEST INFRASTRUCTURE ONLY.  Compiled by oracle/Makefile (target `ref`) together with
// /root/reference/src/simd_dct.cpp and simd_platform.c, from where they lie, into
// oracle/_ref/libsimd_dct_ref.so.  No reference source is copied into this repo; this
// file only declares the reference's externally-linked tier functions
// (simd_dct.cpp:56-67) and its three public entry points (simd_dct.h:29-31) and
// forwards to them, so that Python can pick a tier deterministically.
#include <stddef.h>
#include <stdint.h>

#define TIER(name) \
  void name(const uint8_t *pFrom, uint8_t *pTo, const float *pQuantizeLUT, const size_t sizeX, const size_t sizeY, const size_t startY, const size_t endY)

TIER(simdDCT_EncodeQuantizeReorderStereoBuffer_NoSimd_Float);
TIER(simdDCT_EncodeQuantizeReorderStereoBuffer_SSE41_Float);
TIER(simdDCT_EncodeQuantizeReorderStereoBuffer_SSE2_Float);
TIER(simdDCT_EncodeQuantizeReorderStereoBuffer_SSSE3_Float);
TIER(simdDCT_EncodeQuantizeBuffer_NoSimd_Float);
TIER(simdDCT_EncodeQuantizeBuffer_SSE41_Float);
TIER(simdDCT_EncodeQuantizeBuffer_SSSE3_Float);
TIER(simdDCT_EncodeQuantize32ReorderBuffer_AVX512VL_Float);
TIER(simdDCT_EncodeQuantize32ReorderBuffer_AVX2_Float);

extern "C" void _DetectCPUFeatures(); // simd_platform.c:68 (C linkage via simd_platform.h)

// which: 0 q32/AVX2, 1 q32/AVX-512VL, 2 stereo/SSE4.1, 3 stereo/SSSE3, 4 stereo/SSE2,
//        5 stereo/scalar, 6 encq/SSE4.1, 7 encq/SSSE3, 8 encq/scalar
extern "C" int ref_call_tier(int which, const uint8_t *from, uint8_t *to, const float *lut, size_t sx, size_t sy, size_t y0, size_t y1)
{
  switch (which)
  {
  case 0: simdDCT_EncodeQuantize32ReorderBuffer_AVX2_Float(from, to, lut, sx, sy, y0, y1); return 0;
  case 1: simdDCT_EncodeQuantize32ReorderBuffer_AVX512VL_Float(from, to, lut, sx, sy, y0, y1); return 0;
  case 2: simdDCT_EncodeQuantizeReorderStereoBuffer_SSE41_Float(from, to, lut, sx, sy, y0, y1); return 0;
  case 3: simdDCT_EncodeQuantizeReorderStereoBuffer_SSSE3_Float(from, to, lut, sx, sy, y0, y1); return 0;
  case 4: simdDCT_EncodeQuantizeReorderStereoBuffer_SSE2_Float(from, to, lut, sx, sy, y0, y1); return 0;
  case 5: simdDCT_EncodeQuantizeReorderStereoBuffer_NoSimd_Float(from, to, lut, sx, sy, y0, y1); return 0;
  case 6: simdDCT_EncodeQuantizeBuffer_SSE41_Float(from, to, lut, sx, sy, y0, y1); return 0;
  case 7: simdDCT_EncodeQuantizeBuffer_SSSE3_Float(from, to, lut, sx, sy, y0, y1); return 0;
  case 8: simdDCT_EncodeQuantizeBuffer_NoSimd_Float(from, to, lut, sx, sy, y0, y1); return 0;
  }
  return -1;
}

// The public dispatchers, after the caller-side feature detection main.cpp:449 performs.
enum simdDctResult : int;
simdDctResult simdDCT_EncodeQuantizeBuffer(const uint8_t *, uint8_t *, const float *, const size_t, const size_t, const size_t, const size_t);
simdDctResult simdDCT_EncodeQuantizeReorderStereoBuffer(const uint8_t *, uint8_t *, const float *, const size_t, const size_t, const size_t, const size_t);
simdDctResult simdDCT_EncodeQuantize32ReorderBuffer(const uint8_t *, uint8_t *, const float *, const size_t, const size_t, const size_t, const size_t);

extern "C" void ref_detect_cpu() { _DetectCPUFeatures(); }

// which: 0 q32, 1 stereo, 2 encq
extern "C" int ref_call_public(int which, const uint8_t *from, uint8_t *to, const float *lut, size_t sx, size_t sy, size_t y0, size_t y1)
{
  switch (which)
  {
  case 0: return (int)simdDCT_EncodeQuantize32ReorderBuffer(from, to, lut, sx, sy, y0, y1);
  case 1: return (int)simdDCT_EncodeQuantizeReorderStereoBuffer(from, to, lut, sx, sy, y0, y1);
  case 2: return (int)simdDCT_EncodeQuantizeBuffer(from, to, lut, sx, sy, y0, y1);
  }
  return -1;
}
