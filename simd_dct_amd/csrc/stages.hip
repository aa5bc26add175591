// stages.hip -- the stages either side of the codec core.
//
// The reference's pipeline ends at the reorder store (simd_dct.cpp:2221-2230, :1034-1052) and
// starts from a ready-made plane (main.cpp:475-493 reads one raw file); it has neither of these.
// They are defined against ITU-T T.81 (JPEG): the zig-zag scan of Figure A.6, run/level pairs as
// in F.1.2.2, and JFIF's centred 2x2 chroma siting for 4:2:0.  Integer / byte work, HBM-bound:
// no arithmetic to pin beyond the published tables; the CPU checker generates the scan order by
// walking the anti-diagonals, this file carries it as literals.
//
//   k_scan<SRC, RLE>   after the quantiser: one 8x8 block per lane; the lane's 64 values are
//                      visited in scan order (static register indices), compacted into a
//                      wave-private LDS record [lane][slot] and written out as 16 B per lane.
//   k_split420         before config 3's transform: interleaved 8-bit Y Cb Cr -> three level-shifted
//                      int16 planes, chroma subsampled 2x2 by the rounded box average.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mdct.h"

extern "C" __attribute__((visibility("hidden"))) int mdct_set_error(int code, const char *fmt, ...); // mdct_api.hip

namespace mdct
{

// ITU-T T.81 Figure A.6: natural index (v*8+u) of the k-th coefficient of the zig-zag scan
constexpr int kZigZag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
                             35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

enum { SRC_I16 = 0, SRC_Q32 = 1, SRC_STEREO = 2, SRC_BLOCK = 3 };

struct ScanArgs
{
  const void *src;   // int16 plane (SRC_I16) or the bytes of one of the reference's layouts
  int16_t *levels;   // [block][64]
  uint8_t *runs;     // [block][64]   (RLE only)
  uint8_t *counts;   // [block]       (RLE only)
  size_t pitch;      // SRC_I16: elements;  SRC_STEREO: bytes per coefficient plane
  uint32_t bpr, by0, nblocks; // blocks per (stereo: double) block row, first row, blocks in the launch
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int kWG = 256;
constexpr int kLvRow = 144; // 64 int16 + 16 B: slot 64 (inside the pad) takes the writes of zero coefficients
constexpr int kRnRow = 80;  // 64 u8 + 16 B, same trick
constexpr int kQ32Grp = 520; // staged q32 group: 512 B + 8: a byte read of coefficient c touches banks 2g + 2c + {0,1}, distinct for the 8 groups

template <int SRC, bool RLE>
__global__ __launch_bounds__(kWG) void k_scan(ScanArgs a)
{
  __shared__ __attribute__((aligned(16))) uint8_t lv_all[kWG / 64][64 * kLvRow];
  __shared__ __attribute__((aligned(16))) uint8_t rn_all[kWG / 64][RLE ? 64 * kRnRow : 16];
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t wave_t0 = blockIdx.x * kWG + wave * 64; // first block of the wave within the launch
  if (wave_t0 >= a.nblocks)
    return;
  const uint32_t nvalid = min(64u, a.nblocks - wave_t0);
  const bool valid = lane < nvalid;
  uint8_t *lv = lv_all[wave];
  uint8_t *rn = rn_all[wave];
  const size_t blk0 = (size_t)a.by0 * a.bpr + wave_t0; // absolute index of the wave's first block

  // ---- the lane's 64 values, natural order, as 32-bit integers in registers
  int val[64];
  if constexpr (SRC == SRC_I16)
  {
    const uint32_t t = wave_t0 + (valid ? lane : 0);
    const uint32_t row = t / a.bpr, bx = t - row * a.bpr;
    const int16_t *p = static_cast<const int16_t *>(a.src) + (size_t)(a.by0 + row) * 8 * a.pitch + (size_t)bx * 8;
#pragma unroll
    for (int r = 0; r < 8; r++)
    {
      const u32x4 w = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p + (size_t)r * a.pitch));
#pragma unroll
      for (int j = 0; j < 4; j++)
      {
        val[r * 8 + 2 * j] = (int)(int16_t)(w[j] & 0xFFFF);
        val[r * 8 + 2 * j + 1] = (int)w[j] >> 16;
      }
    }
  }
  else if constexpr (SRC == SRC_Q32)
  { // the wave's 64 blocks are 8 consecutive q32 groups = 4096 contiguous bytes [group][coef][8 blocks]
    const uint8_t *p = static_cast<const uint8_t *>(a.src) + blk0 * 64;
#pragma unroll
    for (int j = 0; j < 4; j++)
    {
      const uint32_t o = (j * 64 + lane) * 16; // byte offset inside the 4 KiB
      if (o < nvalid * 64)
      {
        const u32x4 w = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p + o));
        uint32_t *d = reinterpret_cast<uint32_t *>(lv + o + (kQ32Grp - 512) * (o >> 9));
        d[0] = w.x; d[1] = w.y; d[2] = w.z; d[3] = w.w;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const uint8_t *g = lv + (lane >> 3) * kQ32Grp + (lane & 7);
#pragma unroll
    for (int c = 0; c < 64; c++)
      val[c] = (int)g[c * 8] - 127; // simd_dct.cpp:2224: the stored byte carries a +127 bias
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); // the staging area is reused for the records below
    __builtin_amdgcn_wave_barrier();
  }

  else if constexpr (SRC == SRC_STEREO)
  { // 64 coefficient planes (simd_dct.cpp:1061-1099): coefficient c of stream position p at c * plane + p, so the
    // wave's 64 blocks are 64 consecutive bytes of every plane; bias +127 as in the q32 tier (:1020)
    const uint8_t *p = static_cast<const uint8_t *>(a.src) + blk0 + (valid ? lane : 0);
#pragma unroll
    for (int c = 0; c < 64; c++)
      val[c] = (int)p[(size_t)c * a.pitch] - 127;
  }
  else
  { // SRC_BLOCK (simd_dct.cpp:347-362): 64 contiguous bytes per block, coefficient (v,u) stored at u*8+v, bias +127/255*255
    const uint8_t *p = static_cast<const uint8_t *>(a.src) + (blk0 + (valid ? lane : 0)) * 64;
    uint32_t w[16];
#pragma unroll
    for (int j = 0; j < 4; j++)
    {
      const u32x4 q = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p + j * 16));
      w[4 * j] = q.x; w[4 * j + 1] = q.y; w[4 * j + 2] = q.z; w[4 * j + 3] = q.w;
    }
#pragma unroll
    for (int i = 0; i < 64; i++)
    {
      const int st = (i & 7) * 8 + (i >> 3); // natural v*8+u lives at stored u*8+v
      val[i] = (int)((w[st >> 2] >> (8 * (st & 3))) & 0xFF) - 127;
    }
  }

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
    uint32_t run = 0;
#pragma unroll
    for (int k = 0; k < 64; k++)
    {
      const int c = val[kZigZag[k]];
      const bool nz = c != 0;
      const uint32_t slot = nz ? pos : 64u; // zeros write into the pad
      *reinterpret_cast<int16_t *>(my_lv + slot * 2) = (int16_t)c;
      my_rn[slot] = (uint8_t)run;
      pos += nz ? 1u : 0u;
      run = nz ? 0u : run + 1u;
    }
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
  uint8_t *out_lv = reinterpret_cast<uint8_t *>(a.levels) + blk0 * 128;
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
    uint8_t *out_rn = a.runs + blk0 * 64;
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
      a.counts[blk0 + lane] = (uint8_t)pos;
  }
}

struct SplitArgs
{
  const uint8_t *ycc;
  int16_t *y, *cb, *cr;
  size_t pitch, pitch_y, pitch_c; // bytes / elements / elements
  uint32_t strips, nthreads;      // 8-pixel strips per row pair, threads in the launch
};

typedef unsigned int u32x2_unaligned __attribute__((ext_vector_type(2), aligned(1)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

template <int N>
__device__ __forceinline__ int byte_of(const uint32_t (&w)[6])
{ // byte N of 24 consecutive bytes held in 6 dwords (static index -> one v_bfe_u32)
  return (int)((w[N / 4] >> (8 * (N % 4))) & 0xFF);
}

// one thread: 8 pixels x 2 rows (48 bytes in; 2 x 16 B of Y, 8 B of Cb, 8 B of Cr out)
__global__ __launch_bounds__(kWG) void k_split420(SplitArgs a)
{
  const uint32_t t = blockIdx.x * kWG + threadIdx.x;
  if (t >= a.nthreads)
    return;
  const uint32_t rp = t / a.strips, s = t - rp * a.strips;
  uint32_t w0[6], w1[6];
  const uint8_t *p0 = a.ycc + (size_t)(2 * rp) * a.pitch + (size_t)s * 24, *p1 = p0 + a.pitch;
#pragma unroll
  for (int i = 0; i < 3; i++)
  {
    const u32x2_unaligned v0 = __builtin_nontemporal_load(reinterpret_cast<const u32x2_unaligned *>(p0 + 8 * i));
    const u32x2_unaligned v1 = __builtin_nontemporal_load(reinterpret_cast<const u32x2_unaligned *>(p1 + 8 * i));
    w0[2 * i] = v0.x; w0[2 * i + 1] = v0.y;
    w1[2 * i] = v1.x; w1[2 * i + 1] = v1.y;
  }
  auto pack = [](int lo, int hi) { return ((uint32_t)lo & 0xFFFFu) | ((uint32_t)hi << 16); };
#define Y0(i) (byte_of<3 * (i)>(w0) - 128)
#define Y1(i) (byte_of<3 * (i)>(w1) - 128)
#define C4(k, i) (((byte_of<3 * (2 * (i)) + (k)>(w0) + byte_of<3 * (2 * (i) + 1) + (k)>(w0) + byte_of<3 * (2 * (i)) + (k)>(w1) + byte_of<3 * (2 * (i) + 1) + (k)>(w1) + 2) >> 2) - 128)
  const u32x4 y0 = {pack(Y0(0), Y0(1)), pack(Y0(2), Y0(3)), pack(Y0(4), Y0(5)), pack(Y0(6), Y0(7))};
  const u32x4 y1 = {pack(Y1(0), Y1(1)), pack(Y1(2), Y1(3)), pack(Y1(4), Y1(5)), pack(Y1(6), Y1(7))};
  const u32x2 cb = {pack(C4(1, 0), C4(1, 1)), pack(C4(1, 2), C4(1, 3))};
  const u32x2 cr = {pack(C4(2, 0), C4(2, 1)), pack(C4(2, 2), C4(2, 3))};
#undef Y0
#undef Y1
#undef C4
  int16_t *py = a.y + (size_t)(2 * rp) * a.pitch_y + (size_t)s * 8;
  __builtin_nontemporal_store(y0, reinterpret_cast<u32x4 *>(py));
  __builtin_nontemporal_store(y1, reinterpret_cast<u32x4 *>(py + a.pitch_y));
  __builtin_nontemporal_store(cb, reinterpret_cast<u32x2 *>(a.cb + (size_t)rp * a.pitch_c + (size_t)s * 4));
  __builtin_nontemporal_store(cr, reinterpret_cast<u32x2 *>(a.cr + (size_t)rp * a.pitch_c + (size_t)s * 4));
}

} // namespace mdct

namespace
{

int scan_launch(int src, const void *coef, size_t pitch, size_t sizeX, size_t sizeY, size_t by0, size_t by1, int16_t *levels, uint8_t *runs, uint8_t *counts, void *stream)
{
  if (coef == nullptr || levels == nullptr || (runs != nullptr && counts == nullptr))
    return mdct_set_error(MDCT_INVALID_PARAMETER, "null pointer (runs without counts?)");
  const size_t xmul = src == mdct::SRC_Q32 ? 64 : (src == mdct::SRC_STEREO ? 16 : 8);
  const size_t ymul = src == mdct::SRC_STEREO ? 16 : 8;
  if (sizeX == 0 || sizeX % xmul != 0 || sizeY % ymul != 0)
    return mdct_set_error(MDCT_NOT_SUPPORTED, "plane %zux%zu: width must be a multiple of %zu and height of %zu", sizeX, sizeY, xmul, ymul);
  if ((src == mdct::SRC_I16 && pitch < sizeX) || by0 > by1 || by1 > sizeY / ymul)
    return mdct_set_error(MDCT_INVALID_PARAMETER, "bad pitch or block-row range [%zu,%zu) for %zu rows", by0, by1, sizeY / ymul);
  if (((uintptr_t)levels | (uintptr_t)runs | (src == mdct::SRC_STEREO ? 0 : (uintptr_t)coef) | (src == mdct::SRC_I16 ? pitch * 2 : 0)) & 15)
    return mdct_set_error(MDCT_INVALID_PARAMETER, "coefficient rows and record arrays must be 16-byte aligned");
  if (src == mdct::SRC_STEREO)
    pitch = sizeX * sizeY / 64; // bytes per coefficient plane
  const size_t bpr = (src == mdct::SRC_STEREO ? 2 : 1) * (sizeX / 8), n = bpr * (by1 - by0);
  if (n > 0x7FFFFFFFull)
    return mdct_set_error(MDCT_NOT_SUPPORTED, "more than 2^31 blocks in one call; split the row range");
  if (n == 0)
    return MDCT_SUCCESS;
  mdct::ScanArgs a;
  a.src = coef;
  a.levels = levels;
  a.runs = runs;
  a.counts = counts;
  a.pitch = pitch;
  a.bpr = (uint32_t)bpr;
  a.by0 = (uint32_t)by0;
  a.nblocks = (uint32_t)n;
  const dim3 g((uint32_t)((n + mdct::kWG - 1) / mdct::kWG)), b(mdct::kWG);
  hipStream_t s = (hipStream_t)stream;
#define MDCT_SCAN(SRC) \
  do \
  { \
    if (runs) \
      hipLaunchKernelGGL((mdct::k_scan<SRC, true>), g, b, 0, s, a); \
    else \
      hipLaunchKernelGGL((mdct::k_scan<SRC, false>), g, b, 0, s, a); \
  } while (0)
  switch (src)
  {
  case mdct::SRC_I16: MDCT_SCAN(mdct::SRC_I16); break;
  case mdct::SRC_Q32: MDCT_SCAN(mdct::SRC_Q32); break;
  case mdct::SRC_STEREO: MDCT_SCAN(mdct::SRC_STEREO); break;
  default: MDCT_SCAN(mdct::SRC_BLOCK); break;
  }
#undef MDCT_SCAN
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? MDCT_SUCCESS : mdct_set_error(MDCT_NOT_SUPPORTED, "scan kernel launch: %s", hipGetErrorString(e));
}

} // namespace

extern "C" {

void mdct_zigzag_table(uint8_t *zz64)
{
  for (int k = 0; k < 64; k++)
    zz64[k] = (uint8_t)mdct::kZigZag[k];
}

int mdct_zigzag_rle_i16(const int16_t *coef, size_t pitch, size_t sizeX, size_t sizeY, size_t by0, size_t by1, int16_t *levels, uint8_t *runs, uint8_t *counts, void *stream)
{
  return scan_launch(mdct::SRC_I16, coef, pitch, sizeX, sizeY, by0, by1, levels, runs, counts, stream);
}

int mdct_zigzag_rle_u8(const uint8_t *coef, int layout, size_t sizeX, size_t sizeY, size_t by0, size_t by1, int16_t *levels, uint8_t *runs, uint8_t *counts, void *stream)
{
  const int src = layout == MDCT_LAYOUT_Q32 ? mdct::SRC_Q32 : (layout == MDCT_LAYOUT_STEREO ? mdct::SRC_STEREO : (layout == MDCT_LAYOUT_BLOCK ? mdct::SRC_BLOCK : -1));
  if (src < 0)
    return mdct_set_error(MDCT_NOT_SUPPORTED, "layout %d has no scan (the SSE encq tier stores only half of every block, simd_dct.cpp:1662-1676)", layout);
  return scan_launch(src, coef, sizeX, sizeX, sizeY, by0, by1, levels, runs, counts, stream);
}

int mdct_zigzag_rle_q32(const uint8_t *q32, size_t sizeX, size_t sizeY, size_t by0, size_t by1, int16_t *levels, uint8_t *runs, uint8_t *counts, void *stream)
{
  return mdct_zigzag_rle_u8(q32, MDCT_LAYOUT_Q32, sizeX, sizeY, by0, by1, levels, runs, counts, stream);
}

int mdct_split420_u8(const uint8_t *ycc, size_t pitch, size_t sizeX, size_t sizeY, int16_t *y, int16_t *cb, int16_t *cr, size_t pitch_y, size_t pitch_c, void *stream)
{
  if (ycc == nullptr || y == nullptr || cb == nullptr || cr == nullptr)
    return mdct_set_error(MDCT_INVALID_PARAMETER, "null pointer");
  if (sizeX == 0 || sizeX % 16 != 0 || sizeY % 16 != 0) // both chroma planes must come out as whole 8x8 blocks
    return mdct_set_error(MDCT_NOT_SUPPORTED, "image %zux%zu is not a multiple of 16x16", sizeX, sizeY);
  if (pitch < 3 * sizeX || pitch_y < sizeX || pitch_c < sizeX / 2)
    return mdct_set_error(MDCT_INVALID_PARAMETER, "pitch smaller than a row");
  if (((uintptr_t)y | (pitch_y * 2)) & 15 || ((uintptr_t)cb | (uintptr_t)cr | (pitch_c * 2)) & 7)
    return mdct_set_error(MDCT_INVALID_PARAMETER, "Y rows must be 16-byte and chroma rows 8-byte aligned");
  const size_t strips = sizeX / 8, n = strips * (sizeY / 2);
  if (n > 0x7FFFFFFFull)
    return mdct_set_error(MDCT_NOT_SUPPORTED, "image too large for one call");
  if (n == 0)
    return MDCT_SUCCESS;
  mdct::SplitArgs a;
  a.ycc = ycc;
  a.y = y;
  a.cb = cb;
  a.cr = cr;
  a.pitch = pitch;
  a.pitch_y = pitch_y;
  a.pitch_c = pitch_c;
  a.strips = (uint32_t)strips;
  a.nthreads = (uint32_t)n;
  hipLaunchKernelGGL(mdct::k_split420, dim3((uint32_t)((n + mdct::kWG - 1) / mdct::kWG)), dim3(mdct::kWG), 0, (hipStream_t)stream, a);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? MDCT_SUCCESS : mdct_set_error(MDCT_NOT_SUPPORTED, "split kernel launch: %s", hipGetErrorString(e));
}

} // extern "C"
