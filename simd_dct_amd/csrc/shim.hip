// shim.hip -- the reference's three public functions (simd_dct.h:29-31) on top of the C-ABI.
//
// Translates the reference's call semantics (top-half loop, inclusive endY, tier choice)
// into the engine's half-open block-row ranges, and serves host pointers by staging
// the touched rows through HBM.  Re-entrant: staging buffers are per thread.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstring>

#include "mdct.h"
#include "simd_dct_shim.h"

namespace
{

std::atomic<int> g_max_simd{2};
std::atomic<void *> g_stream{nullptr};
std::atomic<int> g_async{0};

struct Staging
{
  uint8_t *in = nullptr, *out = nullptr; // HBM mirrors of the caller's planes
  size_t in_cap = 0, out_cap = 0;
  // chunk pipeline for host pointers: two pinned bounce buffers per direction, two streams
  uint8_t *pin_in[2] = {nullptr, nullptr}, *pin_out[2] = {nullptr, nullptr};
  size_t pin_cap = 0;
  hipStream_t stream[2] = {nullptr, nullptr};
  int device = -1;
};
thread_local Staging tl_stage;

void release(Staging &s)
{
  if (s.in)
    (void)hipFree(s.in);
  if (s.out)
    (void)hipFree(s.out);
  for (int i = 0; i < 2; i++)
  {
    if (s.pin_in[i])
      (void)hipHostFree(s.pin_in[i]);
    if (s.pin_out[i])
      (void)hipHostFree(s.pin_out[i]);
    if (s.stream[i])
      (void)hipStreamDestroy(s.stream[i]);
  }
  s = Staging();
}

bool reserve_pipeline(Staging &s, size_t chunk_bytes)
{
  for (int i = 0; i < 2; i++)
    if (!s.stream[i] && hipStreamCreateWithFlags(&s.stream[i], hipStreamNonBlocking) != hipSuccess)
      return false;
  if (s.pin_cap >= chunk_bytes)
    return true;
  for (int i = 0; i < 2; i++)
  {
    if (s.pin_in[i])
      (void)hipHostFree(s.pin_in[i]);
    if (s.pin_out[i])
      (void)hipHostFree(s.pin_out[i]);
    s.pin_in[i] = s.pin_out[i] = nullptr;
  }
  s.pin_cap = 0;
  for (int i = 0; i < 2; i++)
    if (hipHostMalloc((void **)&s.pin_in[i], chunk_bytes, hipHostMallocDefault) != hipSuccess || hipHostMalloc((void **)&s.pin_out[i], chunk_bytes, hipHostMallocDefault) != hipSuccess)
    {
      (void)hipGetLastError();
      return false;
    }
  s.pin_cap = chunk_bytes;
  return true;
}

bool reserve(uint8_t *&p, size_t &cap, size_t need)
{
  if (cap >= need)
    return true;
  if (p)
    (void)hipFree(p);
  p = nullptr;
  cap = 0;
  // grow-only with slack so that a loop over similar sizes allocates once
  const size_t want = need + need / 8 + 4096;
  if (hipMalloc((void **)&p, want) != hipSuccess)
  {
    (void)hipGetLastError();
    return false;
  }
  cap = want;
  return true;
}

enum PtrKind { PTR_PAGEABLE, PTR_PINNED, PTR_DEVICE };

PtrKind classify(const void *p)
{
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, p) != hipSuccess)
  {
    (void)hipGetLastError(); // plain malloc memory on older runtimes
    return PTR_PAGEABLE;
  }
  if (attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged)
    return PTR_DEVICE;
  // hipHostMalloc'ed or hipHostRegister'ed (mdct_shim_pin): DMA-able in place, no bounce buffer
  return attr.type == hipMemoryTypeHost ? PTR_PINNED : PTR_PAGEABLE;
}

bool is_device_ptr(const void *p) { return classify(p) == PTR_DEVICE; }

size_t ceil_div(size_t a, size_t b) { return (a + b - 1) / b; }

// Block rows the reference's loop `for (y = 0; y < sizeY/2; y += 8)` processes
// (simd_dct.cpp:2243-2261): step = 16 for the `y*2` tiers, 8 for the scalar encq tier (:375-387).
void ref_range(size_t sizeY, size_t startY, size_t endY, size_t step, size_t *b0, size_t *b1)
{
  const size_t rows = ceil_div(sizeY / 2, 8);
  *b0 = ceil_div(startY, step);
  const size_t last = endY / step + 1;
  *b1 = last < rows ? last : rows;
  if (*b0 > *b1)
    *b0 = *b1;
}

simdDctResult run(const uint8_t *pFrom, uint8_t *pTo, const float *lut, size_t sizeX, size_t sizeY, size_t b0, size_t b1, int layout, int profile)
{
  if (b0 >= b1)
    return sdr_Success;
  const bool dev_in = is_device_ptr(pFrom), dev_out = is_device_ptr(pTo);
  void *stream = g_stream.load();

  if (dev_in && dev_out)
  {
    int r = mdct_fwd_quant_u8(pFrom, pTo, sizeX, lut, sizeX, sizeY, b0, b1, layout, profile, stream);
    if (r == MDCT_SUCCESS && !g_async.load())
      r = mdct_stream_synchronize(stream);
    return (simdDctResult)r;
  }

  // Host path.  Input: the touched pixel rows only.  Output: Q32/BLOCK write whole row
  // strips; STEREO and BLOCK_SSE write scattered bytes, so their destination is first
  // mirrored into HBM to keep untouched bytes untouched.
  Staging &st = tl_stage;
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (st.device != dev)
  {
    release(st);
    st.device = dev;
  }
  const size_t total = sizeX * sizeY;
  const bool scattered = layout == MDCT_LAYOUT_STEREO || layout == MDCT_LAYOUT_BLOCK_SSE;

  // Both pointers on the host and a strip layout: chunked pipeline.  Block-row strips of
  // ~4 MiB ping-pong over two streams, so strip k's kernel and device->host copy overlap strip
  // k+1's host->device copy (PCIe is full duplex); the caller's pageable memory is touched
  // only by plain memcpy to/from pinned bounce buffers.
  if (!dev_in && !dev_out && !scattered)
  {
    const bool pinned_in = classify(pFrom) == PTR_PINNED, pinned_out = classify(pTo) == PTR_PINNED;
    const size_t strip = 8 * sizeX; // bytes per block row, input and output alike
    size_t rows_per_chunk = ((size_t)4 << 20) / strip;
    rows_per_chunk = rows_per_chunk < 1 ? 1 : rows_per_chunk;
    if (reserve(st.in, st.in_cap, total) && reserve(st.out, st.out_cap, total) && reserve_pipeline(st, rows_per_chunk * strip))
    {
      const size_t nchunks = ceil_div(b1 - b0, rows_per_chunk);
      int r = MDCT_SUCCESS;
      auto drain = [&](size_t c) { // chunk c has finished on its stream: hand its output to the caller
        const int sl = (int)(c & 1);
        const size_t r0 = b0 + c * rows_per_chunk, r1 = r0 + rows_per_chunk < b1 ? r0 + rows_per_chunk : b1;
        if (hipStreamSynchronize(st.stream[sl]) != hipSuccess)
          return false;
        if (!pinned_out)
          memcpy(pTo + r0 * strip, st.pin_out[sl], (r1 - r0) * strip);
        return true;
      };
      for (size_t c = 0; c < nchunks && r == MDCT_SUCCESS; c++)
      {
        const int sl = (int)(c & 1);
        if (c >= 2 && !drain(c - 2))
          return sdr_NotSupported;
        const size_t r0 = b0 + c * rows_per_chunk, r1 = r0 + rows_per_chunk < b1 ? r0 + rows_per_chunk : b1;
        const size_t off = r0 * strip, len = (r1 - r0) * strip;
        const uint8_t *h_in = pFrom + off; // pinned caller memory is DMA'd in place
        if (!pinned_in)
        {
          memcpy(st.pin_in[sl], pFrom + off, len);
          h_in = st.pin_in[sl];
        }
        if (hipMemcpyAsync(st.in + off, h_in, len, hipMemcpyHostToDevice, st.stream[sl]) != hipSuccess)
          return sdr_NotSupported;
        r = mdct_fwd_quant_u8(st.in, st.out, sizeX, lut, sizeX, sizeY, r0, r1, layout, profile, st.stream[sl]);
        if (r == MDCT_SUCCESS && hipMemcpyAsync(pinned_out ? pTo + off : st.pin_out[sl], st.out + off, len, hipMemcpyDeviceToHost, st.stream[sl]) != hipSuccess)
          return sdr_NotSupported;
      }
      for (size_t c = nchunks >= 2 ? nchunks - 2 : 0; c < nchunks; c++)
        if (!drain(c) && r == MDCT_SUCCESS)
          return sdr_NotSupported;
      return (simdDctResult)r;
    }
    (void)hipGetLastError(); // could not set the pipeline up: fall through to the plain path
  }

  const uint8_t *d_in = pFrom;
  uint8_t *d_out = pTo;
  hipStream_t hs = (hipStream_t)stream;

  if (!dev_in)
  {
    // STEREO reads both halves of the plane; stage all of it.  Others: rows [b0*8, b1*8).
    const size_t off = layout == MDCT_LAYOUT_STEREO ? 0 : b0 * 8 * sizeX;
    const size_t len = layout == MDCT_LAYOUT_STEREO ? total : (b1 - b0) * 8 * sizeX;
    if (!reserve(st.in, st.in_cap, total))
      return sdr_NotSupported;
    if (hipMemcpyAsync(st.in + off, pFrom + off, len, hipMemcpyHostToDevice, hs) != hipSuccess)
      return sdr_NotSupported;
    d_in = st.in;
  }
  if (!dev_out)
  {
    if (!reserve(st.out, st.out_cap, total))
      return sdr_NotSupported;
    // STEREO over its whole range writes every byte of every coefficient plane: nothing to preserve
    const bool writes_everything = layout == MDCT_LAYOUT_STEREO && b0 == 0 && b1 == sizeY / 16;
    if (scattered && !writes_everything && hipMemcpyAsync(st.out, pTo, total, hipMemcpyHostToDevice, hs) != hipSuccess)
      return sdr_NotSupported;
    d_out = st.out;
  }

  int r = mdct_fwd_quant_u8(d_in, d_out, sizeX, lut, sizeX, sizeY, b0, b1, layout, profile, stream);
  if (r != MDCT_SUCCESS)
    return (simdDctResult)r;

  if (!dev_out)
  {
    const size_t off = scattered ? 0 : b0 * 8 * sizeX;
    const size_t len = scattered ? total : (b1 - b0) * 8 * sizeX;
    if (hipMemcpyAsync(pTo + off, st.out + off, len, hipMemcpyDeviceToHost, hs) != hipSuccess)
      return sdr_NotSupported;
  }
  if (!dev_out || !g_async.load())
    r = mdct_stream_synchronize(stream);
  return (simdDctResult)r;
}

} // namespace

// simd_dct.cpp:113-133
simdDctResult simdDCT_EncodeQuantize32ReorderBuffer(const uint8_t *pFrom, uint8_t *pTo, const float *pQuantizeLUT, const size_t sizeX, const size_t sizeY, const size_t startY, const size_t endY)
{
  if (pFrom == nullptr || pTo == nullptr)
    return sdr_InvalidParameter;
  if ((sizeX & ~(size_t)63) != sizeX || (sizeY & ~(size_t)7) != sizeY)
    return sdr_NotSupported;
  if (g_max_simd.load() < 2)
    return sdr_NotSupported; // :127, no scalar tier exists
  size_t b0, b1;
  ref_range(sizeY, startY, endY, 16, &b0, &b1);
  return run(pFrom, pTo, pQuantizeLUT, sizeX, sizeY, b0, b1, MDCT_LAYOUT_Q32, MDCT_PROFILE_REF_AVX);
}

// simd_dct.cpp:71-91
simdDctResult simdDCT_EncodeQuantizeReorderStereoBuffer(const uint8_t *pFrom, uint8_t *pTo, const float *pQuantizeLUT, const size_t sizeX, const size_t sizeY, const size_t startY, const size_t endY)
{
  if (pFrom == nullptr || pTo == nullptr)
    return sdr_InvalidParameter;
  if ((sizeX & ~(size_t)7) != sizeX || (sizeY & ~(size_t)7) != sizeY)
    return sdr_NotSupported;
  size_t b0, b1;
  ref_range(sizeY, startY, endY, 16, &b0, &b1);
  const int profile = g_max_simd.load() >= 1 ? MDCT_PROFILE_REF_SSE : MDCT_PROFILE_REF_SCALAR;
  return run(pFrom, pTo, pQuantizeLUT, sizeX, sizeY, b0, b1, MDCT_LAYOUT_STEREO, profile);
}

// simd_dct.cpp:93-111
simdDctResult simdDCT_EncodeQuantizeBuffer(const uint8_t *pFrom, uint8_t *pTo, const float *pQuantizeLUT, const size_t sizeX, const size_t sizeY, const size_t startY, const size_t endY)
{
  if (pFrom == nullptr || pTo == nullptr)
    return sdr_InvalidParameter;
  if ((sizeX & ~(size_t)7) != sizeX || (sizeY & ~(size_t)7) != sizeY)
    return sdr_NotSupported;
  size_t b0, b1;
  if (g_max_simd.load() >= 1)
  {
    ref_range(sizeY, startY, endY, 16, &b0, &b1);
    return run(pFrom, pTo, pQuantizeLUT, sizeX, sizeY, b0, b1, MDCT_LAYOUT_BLOCK_SSE, MDCT_PROFILE_REF_SSE);
  }
  ref_range(sizeY, startY, endY, 8, &b0, &b1);
  return run(pFrom, pTo, pQuantizeLUT, sizeX, sizeY, b0, b1, MDCT_LAYOUT_BLOCK, MDCT_PROFILE_REF_SCALAR);
}

extern "C" {

void mdct_shim_set_max_simd(int level) { g_max_simd.store(level < 0 ? 0 : (level > 2 ? 2 : level)); }
void mdct_shim_set_stream(void *stream) { g_stream.store(stream); }
void mdct_shim_set_async(int enabled) { g_async.store(enabled != 0); }
void mdct_shim_release(void) { release(tl_stage); }

int mdct_shim_pin(void *p, size_t bytes)
{
  if (p == nullptr || bytes == 0)
    return MDCT_INVALID_PARAMETER;
  if (hipHostRegister(p, bytes, hipHostRegisterDefault) != hipSuccess)
  {
    (void)hipGetLastError();
    return MDCT_NOT_SUPPORTED;
  }
  return MDCT_SUCCESS;
}

int mdct_shim_unpin(void *p)
{
  if (p == nullptr)
    return MDCT_INVALID_PARAMETER;
  if (hipHostUnregister(p) != hipSuccess)
  {
    (void)hipGetLastError();
    return MDCT_NOT_SUPPORTED;
  }
  return MDCT_SUCCESS;
}

// C handles onto the C++-linkage functions above, for FFI callers (ctypes / cgo / JNI)
// that cannot spell Itanium-mangled names.  which: 0 q32, 1 stereo, 2 encq.
int mdct_shim_call(int which, const uint8_t *pFrom, uint8_t *pTo, const float *lut, size_t sizeX, size_t sizeY, size_t startY, size_t endY)
{
  switch (which)
  {
  case 0: return (int)simdDCT_EncodeQuantize32ReorderBuffer(pFrom, pTo, lut, sizeX, sizeY, startY, endY);
  case 1: return (int)simdDCT_EncodeQuantizeReorderStereoBuffer(pFrom, pTo, lut, sizeX, sizeY, startY, endY);
  case 2: return (int)simdDCT_EncodeQuantizeBuffer(pFrom, pTo, lut, sizeX, sizeY, startY, endY);
  }
  return MDCT_INVALID_PARAMETER;
}

} // extern "C"
