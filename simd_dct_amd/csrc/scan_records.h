// scan_records.h -- the back half of the zig-zag / run-level stage, shared by k_scan (stages.hip) and the fused
// pixels -> records kernel (mdct_kernels.hip).  Device code only; no reference counterpart (ITU-T T.81 Figure A.6,
// F.1.2.2: see stages.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mdct
{

typedef unsigned int scan_u32x4 __attribute__((ext_vector_type(4)));

// ITU-T T.81 Figure A.6: natural index (v*8+u) of the k-th coefficient of the zig-zag scan
constexpr int kZigZag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
                             35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

constexpr int kLvRow = 144; // 64 int16 + 16 B of pad (slot 64 takes the write that clears the run after the last pair)
constexpr int kRnRow = 80;  // 64 u8 + 16 B, same trick

// The lane's 64 values (natural order v*8+u, 32-bit integers in registers) -> its record in the wave-private LDS areas
// lv (64 x kLvRow bytes) and rn (64 x kRnRow), then the wave's 64 records out as 16 B per lane: the wave's blocks are
// blk0 .. blk0 + nvalid - 1 of the record arrays, lane = block.  All 64 lanes must call (wave barrier inside).
template <bool RLE>
__device__ __forceinline__ void scan_emit(const int (&val)[64], uint8_t *lv, uint8_t *rn, uint32_t lane, uint32_t nvalid, bool valid, size_t blk0, int16_t *levels, uint8_t *runs,
                                          uint8_t *counts)
{
  typedef scan_u32x4 u32x4;
  // ---- scan order; RLE: compact the non-zero levels to the front of the lane's record
  uint8_t *my_lv = lv + lane * kLvRow;
  uint32_t pos = 0;
  if constexpr (RLE)
  {
    uint8_t *my_rn = rn + lane * kRnRow;
    const u32x4 z = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < kLvRow / 16; i++)
      *reinterpret_cast<u32x4 *>(my_lv + i * 16) = z;
#pragma unroll
    for (int i = 0; i < kRnRow / 16; i++)
      *reinterpret_cast<u32x4 *>(my_rn + i * 16) = z;
    // every coefficient writes (level, run) at the current position and only the non-zero ones advance it: a zero's write is
    // overwritten by the next pair; what the last zeros leave at the final position is level 0 and a run that is cleared below
    // (no per-coefficient select of the slot)
    uint32_t run = 0;
#pragma unroll
    for (int k = 0; k < 64; k++)
    {
      const int c = val[kZigZag[k]];
      const bool nz = c != 0;
      *reinterpret_cast<int16_t *>(my_lv + pos * 2) = (int16_t)c; // pos <= 63 while coefficients remain; slot 64 is inside the pad
      my_rn[pos] = (uint8_t)run;
      pos += nz ? 1u : 0u;
      run = nz ? 0u : run + 1u;
    }
    my_rn[pos] = 0; // trailing zeros left their run at [pos]: clear it -- runs beyond the last pair are 0 in the record format (pos == 64 lands in the pad)
  }
  else
  {
#pragma unroll
    for (int k = 0; k < 64; k += 2)
      *reinterpret_cast<uint32_t *>(my_lv + k * 2) = ((uint32_t)val[kZigZag[k]] & 0xFFFFu) | ((uint32_t)val[kZigZag[k + 1]] << 16);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  // ---- records out: the wave's 64 x 128 B of levels (and 64 x 64 B of runs) are contiguous; 16 B per lane per store
  uint8_t *out_lv = reinterpret_cast<uint8_t *>(levels) + blk0 * 128;
#pragma unroll
  for (int j = 0; j < 8; j++)
  {
    const uint32_t b = j * 8 + (lane >> 3);
    if (b < nvalid)
    {
      const u32x4 w = *reinterpret_cast<const u32x4 *>(lv + b * kLvRow + (lane & 7) * 16);
      __builtin_nontemporal_store(w, reinterpret_cast<u32x4 *>(out_lv + (j * 64 + lane) * 16));
    }
  }
  if constexpr (RLE)
  {
    uint8_t *out_rn = runs + blk0 * 64;
#pragma unroll
    for (int j = 0; j < 4; j++)
    {
      const uint32_t b = j * 16 + (lane >> 2);
      if (b < nvalid)
      {
        const u32x4 w = *reinterpret_cast<const u32x4 *>(rn + b * kRnRow + (lane & 3) * 16);
        __builtin_nontemporal_store(w, reinterpret_cast<u32x4 *>(out_rn + (j * 64 + lane) * 16));
      }
    }
    if (valid)
      counts[blk0 + lane] = (uint8_t)pos;
  }
}

} // namespace mdct
