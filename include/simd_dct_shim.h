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

/* simd_dct.h:7-20: parameter decorations and the status tests callers of the reference use */
#ifndef IN
#define IN
#endif
#ifndef OUT
#define OUT
#endif
#ifndef IN_OUT
#define IN_OUT IN OUT
#endif
#ifndef _SUCCEEDED
#define _SUCCEEDED(errorCode) (sdr_Success == (errorCode))
#endif
#ifndef _FAILED
#define _FAILED(errorCode) (!(_SUCCEEDED(errorCode)))
#endif

#ifdef __cplusplus

/* simd_dct.h:22-27 */
enum simdDctResult
{
  sdr_Success,
  sdr_InvalidParameter,
  sdr_NotSupported,
};

/* Deviations from the reference, all on inputs where the reference reads or writes out of bounds
 * (DESIGN.md 1): the stereo function needs sizeY % 16 == 0 and the SSE tiers sizeX % 16 == 0 -- the
 * reference walks 16 px / two eyes at a time and returns sdr_Success after over-reading
 * (simd_dct.cpp:945, :1591); here such planes return sdr_NotSupported.  With the tier cap at SSE4.1
 * the reference's q32 would run its lane-mis-packed SSE4.1 variant (:2267-2539); here: sdr_NotSupported. */

/* simd_dct.h:29 -> tiers simd_dct.cpp:1540 (SSE4.1, default) / :300 (scalar) */
simdDctResult simdDCT_EncodeQuantizeBuffer(IN const uint8_t *pFrom, OUT uint8_t *pTo, IN const float *pQuantizeLUT, const size_t sizeX, const size_t sizeY, const size_t startY, const size_t endY);
/* simd_dct.h:30 -> tiers simd_dct.cpp:896 (SSE4.1, default) / :177 (scalar) */
simdDctResult simdDCT_EncodeQuantizeReorderStereoBuffer(IN const uint8_t *pFrom, OUT uint8_t *pTo, IN const float *pQuantizeLUT, const size_t sizeX, const size_t sizeY, const size_t startY, const size_t endY);
/* simd_dct.h:31 -> tier simd_dct.cpp:2064 (AVX2 == AVX-512VL) */
simdDctResult simdDCT_EncodeQuantize32ReorderBuffer(IN const uint8_t *pFrom, OUT uint8_t *pTo, IN const float *pQuantizeLUT, const size_t sizeX, const size_t sizeY, const size_t startY, const size_t endY);

extern "C" {
#endif

/* Counterpart of the reference's mutable CPU-flag globals / `--max-simd` (main.cpp:87-97, :283-438):
 * the highest reference tier the shim reproduces, process-wide like those globals.
 *   level            q32 (simd_dct.cpp:120-127)   stereo (:78-85)        encq (:100-105)
 *   MDCT_SIMD_NONE   sdr_NotSupported             scalar :177            scalar :300
 *   MDCT_SIMD_SSE2   sdr_NotSupported             SSE2 :1106             scalar :300 (no SSE2 tier exists)
 *   MDCT_SIMD_SSSE3  sdr_NotSupported             SSSE3 :1330            SSSE3 :1707
 *   MDCT_SIMD_SSE41  sdr_NotSupported (see above) SSE4.1 :896            SSE4.1 :1540
 *   MDCT_SIMD_AVX2   AVX2 :2064 == AVX-512VL      SSE4.1                 SSE4.1
 * (the SSE tiers of one function write identical bytes.)  Default when never called, or called with
 * a negative level: if the program also links the reference's simd_platform.c, the shim follows its
 * flag globals (sse2Supported ... avx512VLSupported, simd_platform.h:21-46) exactly as the reference's
 * dispatchers do -- so the reference's own main.cpp, relinked, keeps its `--max-simd` behaviour and
 * gets the scalar tiers until it has called _DetectCPUFeatures(); otherwise MDCT_SIMD_AVX2. */
enum
{
  MDCT_SIMD_NONE = 0,
  MDCT_SIMD_SSE2 = 1,
  MDCT_SIMD_SSSE3 = 2,
  MDCT_SIMD_SSE41 = 3,
  MDCT_SIMD_AVX2 = 4
};
void mdct_shim_set_max_simd(int level);
/* the level in effect right now (after flag-following) */
int mdct_shim_get_max_simd(void);
/* stream (hipStream_t as void*) used for device-pointer calls, and whether they return before
 * completion.  Per HOST THREAD (every thread that calls the three functions has its own setting,
 * default: null stream, synchronous).  Host-pointer calls are synchronous and use the thread's
 * internal copy streams; for pageable q32 / scalar-encq calls larger than one ~4 MiB strip the
 * calling thread also keeps three helper threads that share the copies between the caller's memory
 * and the pinned bounce buffers (they sleep between calls and end with the thread or with
 * mdct_shim_release()). */
void mdct_shim_set_stream(void *stream);
void mdct_shim_set_async(int enabled);
/* frees the calling thread's staging buffers and ends its helper threads now (both also happen when the thread exits) */
void mdct_shim_release(void);
/* Optional: pay the one-time costs of the calling thread's first host-pointer call NOW instead of inside it -- HIP and
 * device initialisation, code-object load, the device mirrors of a plane of `plane_bytes` (sizeX * sizeY) in each
 * direction, the pinned bounce buffers, streams and helper threads of the chunk pipeline.  A caller that times calls
 * the way main.cpp:510-523 does otherwise sees one call of ~0.2 s among calls of ~1.5 ms.  Returns 0 / 1 / 2. */
int mdct_shim_warmup(size_t plane_bytes);
/* Optional: page-lock a caller-owned host buffer that is reused across calls (hipHostRegister).
 * Host-pointer calls then DMA to/from it in place instead of bouncing through the shim's pinned
 * buffers with memcpy.  Unpin before freeing it.  Returns 0 / 1 / 2 like every entry point. */
int mdct_shim_pin(void *p, size_t bytes);
int mdct_shim_unpin(void *p);
/* The same without a source change -- OPT-IN through the environment, read once per process:
 *   MDCT_SHIM_AUTOPIN=1        a host plane passed to the three functions for the THIRD time (same base pointer; the longest extent a
 *                              call has touched) is page-locked in place by the shim and DMA'd from / to directly from then on: what the
 *                              reference's harness needs, which reuses its two buffers for every run (main.cpp:510-523).  At most 16
 *                              ranges per process, least recently used released first; all released by mdct_shim_release(), by a calling
 *                              thread's exit and at process exit.  THE CALLER'S PART OF THE BARGAIN: a buffer it has passed three times is
 *                              not freed before one of those -- a range freed and re-allocated at the same address while registered would be
 *                              read / written through its old pages.  That is why this is not the default.
 *   MDCT_SHIM_EVENT_WAIT=block the host threads of the pipeline sleep in their event waits (hipEventBlockingSync) instead of spinning:
 *                              frees up to 3-4 cores per calling thread, costs 5-12 % of the call (profiles/r06_host_pointer_*_ab.log). */
/* C-linkage handle onto the three C++-linkage functions above for FFI callers that cannot
 * spell mangled names (ctypes, cgo, JNI).  which: 0 = ...32ReorderBuffer (simd_dct.h:31),
 * 1 = ...ReorderStereoBuffer (simd_dct.h:30), 2 = ...EncodeQuantizeBuffer (simd_dct.h:29).
 * Returns the simdDctResult value. */
int mdct_shim_call(int which, const uint8_t *pFrom, uint8_t *pTo, const float *pQuantizeLUT,
                   size_t sizeX, size_t sizeY, size_t startY, size_t endY);
/* the same with the stream / async choice passed per call instead of per thread */
int mdct_shim_call_on(int which, const uint8_t *pFrom, uint8_t *pTo, const float *pQuantizeLUT,
                      size_t sizeX, size_t sizeY, size_t startY, size_t endY, void *stream, int async);

#ifdef __cplusplus
}
#endif
#endif
