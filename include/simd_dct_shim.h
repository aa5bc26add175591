/*
 * simd_dct_shim.h -- the reference's public API, served by the MI355X engine.
 *
 * Host code written against rainerzufalldererste/simd_dct's `simd_dct.h` relinks against
 * libmdct_hip.so unchanged: the three functions below have the reference's names,
 * C++ linkage (the reference header has no extern "C"; Itanium-mangled
 * _Z37simdDCT_EncodeQuantize32ReorderBufferPKhPhPKfmmmm etc.), argument order, result
 * enum and row-range semantics (simd_dct.h:22-31).  A project that still has the
 * reference's own header on its include path can keep using that header; this one
 * exists so the engine is usable without it.
 *
 * Pointers may be HOST pointers (the reference's only mode; the shim stages through HBM
 * and returns when the output is complete) or DEVICE pointers (zero-copy; the call
 * returns after the kernel has finished unless mdct_shim_set_async(1)).
 *
 * Reference semantics reproduced: only block rows y < sizeY/2 are
 * processed, a row is processed iff startY <= 2*y <= endY (encq scalar tier:
 * startY <= y <= endY), bytes the reference leaves untouched stay untouched.
 */
#ifndef SIMD_DCT_SHIM_H
#define SIMD_DCT_SHIM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus

/* simd_dct.h:22-27 */
enum simdDctResult
{
  sdr_Success,
  sdr_InvalidParameter,
  sdr_NotSupported,
};

/* simd_dct.h:29 -> tiers simd_dct.cpp:1540 (SSE4.1, default) / :300 (scalar) */
simdDctResult simdDCT_EncodeQuantizeBuffer(const uint8_t *pFrom, uint8_t *pTo, const float *pQuantizeLUT, const size_t sizeX, const size_t sizeY, const size_t startY, const size_t endY);
/* simd_dct.h:30 -> tiers simd_dct.cpp:896 (SSE4.1, default) / :177 (scalar) */
simdDctResult simdDCT_EncodeQuantizeReorderStereoBuffer(const uint8_t *pFrom, uint8_t *pTo, const float *pQuantizeLUT, const size_t sizeX, const size_t sizeY, const size_t startY, const size_t endY);
/* simd_dct.h:31 -> tier simd_dct.cpp:2064 (AVX2 == AVX-512VL) */
simdDctResult simdDCT_EncodeQuantize32ReorderBuffer(const uint8_t *pFrom, uint8_t *pTo, const float *pQuantizeLUT, const size_t sizeX, const size_t sizeY, const size_t startY, const size_t endY);

extern "C" {
#endif

/* Counterpart of the reference's mutable CPU-flag globals / `--max-simd` (main.cpp:283-438):
 * which reference tier the shim reproduces.  0 = none: scalar tiers, and q32 returns
 * sdr_NotSupported exactly as simd_dct.cpp:127 does; 1 = SSE: stereo/encq SSE tiers, q32
 * not supported (the reference's SSE4.1 q32 variant mis-packs lanes and is not reproduced);
 * 2 = AVX2 (default): what an AVX2 host runs after _DetectCPUFeatures(). */
void mdct_shim_set_max_simd(int level);
/* stream (hipStream_t as void*) used for device-pointer calls, and whether they return
 * before completion.  Per process. */
void mdct_shim_set_stream(void *stream);
void mdct_shim_set_async(int enabled);
/* frees the calling thread's staging buffers */
void mdct_shim_release(void);
/* Optional: page-lock a caller-owned host buffer that is reused across calls (hipHostRegister).
 * Host-pointer calls then DMA to/from it in place instead of bouncing through the shim's pinned
 * buffers with memcpy.  Unpin before freeing it.  Returns 0 / 1 / 2 like every entry point. */
int mdct_shim_pin(void *p, size_t bytes);
int mdct_shim_unpin(void *p);
/* C-linkage handle onto the three C++-linkage functions above for FFI callers that cannot
 * spell mangled names (ctypes, cgo, JNI).  which: 0 = ...32ReorderBuffer (simd_dct.h:31),
 * 1 = ...ReorderStereoBuffer (simd_dct.h:30), 2 = ...EncodeQuantizeBuffer (simd_dct.h:29).
 * Returns the simdDctResult value. */
int mdct_shim_call(int which, const uint8_t *pFrom, uint8_t *pTo, const float *pQuantizeLUT,
                   size_t sizeX, size_t sizeY, size_t startY, size_t endY);

#ifdef __cplusplus
}
#endif
#endif
