// mdct_kernels.h -- launch interface between the C-ABI layer (mdct_api.hip) and the
// gfx950 kernels (mdct_kernels.hip).  Internal; the public boundary is include/mdct.h.
#ifndef MDCT_KERNELS_H
#define MDCT_KERNELS_H

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "mdct.h"

namespace mdct
{

enum { MODE_FWD = 0, MODE_INV = 1, MODE_ROUNDTRIP = 2 };

constexpr int kMaxPlanes = 4;

// 64 quantiser multipliers, passed BY VALUE in the kernarg segment so the kernel reads
// them with scalar loads (wave-uniform, no VGPR or LDS cost).
struct QuantTable
{
  float q[64];
};

struct LutPair
{
  float rq[64];  // 1.0f / lut[i]  (forward quantise multiplier)
  float lut[64]; // dequantise multiplier
};

struct U8Args
{
  const uint8_t *from;
  uint8_t *to;
  QuantTable qt;
  size_t pitch;        // input row pitch, bytes
  size_t sizeX;        // plane width, bytes (output addressing)
  size_t eye_offset;   // STEREO: byte offset of the second image
  size_t plane_stride; // STEREO: bytes per coefficient plane (sizeX*sizeY/64)
  uint32_t bpr;        // blocks per block row (sizeX/8)
  uint32_t by0;        // first block row of the launch
  uint32_t by_last;    // BLOCK_SSE: last block row of the launch
  uint32_t nblocks;    // blocks in the launch
  uint32_t aligned8;   // input rows 8-byte aligned
  uint32_t spill_ok;   // BLOCK_SSE: trailing spill stays inside the buffer
};

struct I16Args
{
  const int16_t *from;
  int16_t *to;
  LutPair lp;
  size_t pitch_in, pitch_out; // elements
  uint32_t bpr, by0, nblocks;
};

struct F32Args
{
  const float *from;
  float *to;
  size_t pitch_in, pitch_out; // elements
  uint32_t bpr, by0, nblocks;
};

struct PlaneBatchArgs
{
  const int16_t *from[kMaxPlanes];
  int16_t *to[kMaxPlanes];
  size_t pitch_in[kMaxPlanes], pitch_out[kMaxPlanes];
  uint32_t bpr[kMaxPlanes];
  uint32_t nblk[kMaxPlanes];       // real block count of each plane
  uint32_t prefix[kMaxPlanes + 1]; // exclusive scan of the block counts padded to whole waves
  uint32_t has_lut[kMaxPlanes];    // 0: plain fwd->inv, 1: quantise/dequantise in between
  int n;
  LutPair lp[kMaxPlanes];
};

hipError_t launch_fwd_quant_u8(const U8Args &a, int layout, int profile, bool safe, hipStream_t s);
hipError_t launch_i16(const I16Args &a, int mode, bool has_lut, hipStream_t s);
hipError_t launch_i16_planes(const PlaneBatchArgs &a, hipStream_t s);
hipError_t launch_f32(const F32Args &a, int mode, hipStream_t s);
hipError_t launch_stream_copy(const void *from, void *to, size_t bytes, int cus, hipStream_t s);

} // namespace mdct
#endif
