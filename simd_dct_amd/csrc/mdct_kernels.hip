// mdct_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the 8x8 block-DCT engine.
//
// Mapping (all kernels): ONE 8x8 BLOCK PER LANE, a wave64 carries 64 consecutive blocks.
// A lane owns its whole block in VGPRs, so the row pass, the column pass and the
// row<->column "transpose" are pure register renaming; cross-lane traffic exists only in
// the OUTPUT REORDER of the interleaved layouts and goes through wave-private LDS.
// There is no dense contraction anywhere: no MFMA.
//
// Bit-exactness contract (u8 path): IEEE binary32, one rounding per written operation,
// the exact association of the reference tier being reproduced, NO FMA.  This file is
// compiled with -ffp-contract=off and additionally pins contraction off below.
//
// Reference lines restated (rainerzufalldererste/simd_dct, src/simd_dct.cpp):
//   1-D kernels   K_AVX :2158-2184   K_SSE :434-654   K_TRUE :138-172
//   block drivers B1 :2064-2262  B2 :896-1103  B3 :1540-1704  B4 :177-298  B5 :300-395
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mdct_kernels.h"

#pragma clang fp contract(off)

namespace mdct
{

// simd_dct.cpp:140-146
__device__ constexpr float kCa = 1.3870398453221474618216191915664f;
__device__ constexpr float kCb = 1.3065629648763765278566431734272f;
__device__ constexpr float kCc = 1.1758756024193587169744671046113f;
__device__ constexpr float kCd = 0.78569495838710218127789736765722f;
__device__ constexpr float kCe = 0.54119610014619698439972320536639f;
__device__ constexpr float kCf = 0.27589937928294301233595756366937f;
__device__ constexpr float kCn = 0.35355339059327376220042218105242f;

enum { K_AVX = 0, K_SSE = 1, K_TRUE = 2, K_OWN = 3 };

// ---------------------------------------------------------------------------------------
// 1-D 8-point forward kernel on eight registers.
// ---------------------------------------------------------------------------------------
template <int K>
__device__ __forceinline__ void dct8(float &p0, float &p1, float &p2, float &p3, float &p4, float &p5, float &p6, float &p7)
{
  const float x07p = p0 + p7, x16p = p1 + p6, x25p = p2 + p5, x34p = p3 + p4;
  const float x07m = p0 - p7, x61m = p6 - p1, x25m = p2 - p5, x43m = p4 - p3;
  const float pp = x07p + x34p, pm = x07p - x34p;
  const float qp = x16p + x25p, qm = x16p - x25p;

  const float o0 = pp + qp;
  const float o4 = pp - qp;
  float o1, o2, o3, o5, o6, o7;

  if constexpr (K == K_TRUE)
  { // left-to-right association (:163-171)
    o2 = kCb * pm + kCe * qm;
    o6 = kCe * pm - kCb * qm;
    o1 = ((kCa * x07m - kCc * x61m) + kCd * x25m) - kCf * x43m;
    o3 = ((kCc * x07m + kCf * x61m) - kCa * x25m) + kCd * x43m;
    o5 = ((kCd * x07m + kCa * x61m) + kCf * x25m) - kCc * x43m;
    o7 = ((kCf * x07m + kCd * x61m) + kCc * x25m) + kCa * x43m;
  }
  else
  { // pairwise association; a + (-b) == a - b and (-c)*x == -(c*x) exactly in IEEE,
    // so only the k=1 (K_SSE, :550) and k=3 (K_AVX, :2181) sign quirks differ.
    o2 = (kCb * pm) + (kCe * qm);
    o6 = (kCe * pm) - (kCb * qm);
    const float t1 = (kCa * x07m) - (kCc * x61m);
    const float t3 = (kCc * x07m) + (kCf * x61m);
    const float t5 = (kCd * x07m) + (kCa * x61m);
    const float t7 = (kCf * x07m) + (kCd * x61m);
    if constexpr (K == K_SSE)
    {
      o1 = t1 + ((kCd * x25m) + (kCf * x43m)); // quirk: +Cf
      o3 = t3 + ((kCd * x43m) - (kCa * x25m)); // ((-Ca)*x25m) + (Cd*x43m)
    }
    else if constexpr (K == K_AVX)
    {
      o1 = t1 + ((kCd * x25m) - (kCf * x43m));
      o3 = t3 - ((kCa * x25m) + (kCd * x43m)); // quirk: -Cd
    }
    else
    {
      o1 = t1 + ((kCd * x25m) - (kCf * x43m));
      o3 = t3 - ((kCa * x25m) - (kCd * x43m));
    }
    o5 = t5 + ((kCf * x25m) - (kCc * x43m));
    o7 = t7 + ((kCc * x25m) + (kCa * x43m));
  }

  p0 = kCn * o0; p1 = kCn * o1; p2 = kCn * o2; p3 = kCn * o3;
  p4 = kCn * o4; p5 = kCn * o5; p6 = kCn * o6; p7 = kCn * o7;
}

// Inverse of K_OWN (transposed flow graph).
__device__ __forceinline__ void idct8(float &p0, float &p1, float &p2, float &p3, float &p4, float &p5, float &p6, float &p7)
{
  const float a0 = p0 + p4, a1 = p0 - p4;
  const float b0 = (kCb * p2) + (kCe * p6);
  const float b1 = (kCe * p2) - (kCb * p6);
  const float e0 = a0 + b0, e1 = a1 + b1, e2 = a1 - b1, e3 = a0 - b0;

  const float d0 = ((kCa * p1) + (kCc * p3)) + ((kCd * p5) + (kCf * p7));
  const float d1 = ((kCc * p1) - (kCf * p3)) - ((kCa * p5) + (kCd * p7));
  const float d2 = ((kCd * p1) - (kCa * p3)) + ((kCf * p5) + (kCc * p7));
  const float d3 = ((kCf * p1) - (kCd * p3)) + ((kCc * p5) - (kCa * p7));

  p0 = kCn * (e0 + d0); p7 = kCn * (e0 - d0);
  p1 = kCn * (e1 + d1); p6 = kCn * (e1 - d1);
  p2 = kCn * (e2 + d2); p5 = kCn * (e2 - d2);
  p3 = kCn * (e3 + d3); p4 = kCn * (e3 - d3);
}

template <int K>
__device__ __forceinline__ void pass_rows(float (&b)[8][8])
{
#pragma unroll
  for (int r = 0; r < 8; r++)
    dct8<K>(b[r][0], b[r][1], b[r][2], b[r][3], b[r][4], b[r][5], b[r][6], b[r][7]);
}

template <int K>
__device__ __forceinline__ void pass_cols(float (&b)[8][8])
{
#pragma unroll
  for (int c = 0; c < 8; c++)
    dct8<K>(b[0][c], b[1][c], b[2][c], b[3][c], b[4][c], b[5][c], b[6][c], b[7][c]);
}

__device__ __forceinline__ void ipass_rows(float (&b)[8][8])
{
#pragma unroll
  for (int r = 0; r < 8; r++)
    idct8(b[r][0], b[r][1], b[r][2], b[r][3], b[r][4], b[r][5], b[r][6], b[r][7]);
}

__device__ __forceinline__ void ipass_cols(float (&b)[8][8])
{
#pragma unroll
  for (int c = 0; c < 8; c++)
    idct8(b[0][c], b[1][c], b[2][c], b[3][c], b[4][c], b[5][c], b[6][c], b[7][c]);
}

// ---------------------------------------------------------------------------------------
// Quantisers.  x86 cvtps_epi32 = RNE, "integer indefinite" 0x80000000 when out of range
// or NaN.  CDNA v_cvt_i32_f32 saturates and maps NaN to 0, so the result after the
// reference's bias+clamp differs only for NaN (x86: 0x80000000 -> clamp -> 0) and, for
// the float-bias tiers, for +overflow; SAFE handles exactly those (reachable only with
// non-finite or absurd quantisers -- the host picks SAFE when a table entry is not finite
// or exceeds 2^20, see mdct_api.hip).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ int32_t cvt_rne(float v) { return (int32_t)__builtin_rintf(v); } // v_rndne_f32 + v_cvt_i32_f32

template <bool SAFE>
__device__ __forceinline__ int32_t cvtps_epi32(float v)
{
  if constexpr (SAFE)
  {
    if (!(__builtin_fabsf(v) < 2147483648.0f))
      return INT32_MIN;
  }
  return cvt_rne(v);
}

__device__ __forceinline__ int32_t clamp255(int32_t v) { return min(max(v, 0), 255); } // v_med3_i32

// B1 :2224  clamp(127 + rne(f*q))   (wrapping int32 add, like _mm256_add_epi32)
template <bool SAFE>
__device__ __forceinline__ uint32_t quant_avx(float f, float q)
{
  const int32_t r = cvtps_epi32<SAFE>(f * q);
  return (uint32_t)clamp255((int32_t)((uint32_t)r + 127u));
}

// B2/B3 :1020  clamp(rne(f*q + 127.0f))
template <bool SAFE>
__device__ __forceinline__ uint32_t quant_sse(float f, float q)
{
  return (uint32_t)clamp255(cvtps_epi32<SAFE>((f * q) + 127.0f));
}

// B4/B5 :245, :362  (uint8_t)roundf(_clamp(f*qs + 127/255, 0, 1) * 255)
__device__ __forceinline__ uint32_t quant_scalar(float f, float qs)
{
  float v = (f * qs) + (127.0f / 255.0f);
  v = v > 0.f ? (v < 1.f ? v : 1.f) : 0.f;
  return (uint32_t)roundf(v * 255.f);
}

// ---------------------------------------------------------------------------------------
// u8 forward + quantise + reorder.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ uint2 load8(const uint8_t *p, bool aligned)
{
  if (aligned)
    return *reinterpret_cast<const uint2 *>(p);
  uint2 v;
  v.x = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
  v.y = (uint32_t)p[4] | ((uint32_t)p[5] << 8) | ((uint32_t)p[6] << 16) | ((uint32_t)p[7] << 24);
  return v;
}

template <int PROFILE>
__device__ __forceinline__ float px_to_float(uint32_t px)
{
  const float f = (float)px; // v_cvt_f32_ubyteN
  if constexpr (PROFILE == MDCT_PROFILE_REF_AVX)
    return f; // :2143, raw 0..255
  else if constexpr (PROFILE == MDCT_PROFILE_REF_SSE)
    return (1.f / (float)0xFF) * f; // :949
  else
    return f / 255.f; // :222, :343 (true division)
}

// Loads the lane's block, runs both passes in the profile's order and returns the 64
// quantised bytes as int values q[v][u] (natural index) for the AVX/stereo layouts or
// q[u][v]-transposed-stored semantics handled by the caller.
template <int PROFILE, int LAYOUT, bool SAFE>
__device__ __forceinline__ void encode_block(const uint8_t *src, size_t pitch, bool aligned, const QuantTable &qt, uint32_t (&out)[64])
{
  float b[8][8];
#pragma unroll
  for (int r = 0; r < 8; r++)
  {
    const uint2 v = load8(src + (size_t)r * pitch, aligned);
    b[r][0] = px_to_float<PROFILE>(v.x & 0xFF);
    b[r][1] = px_to_float<PROFILE>((v.x >> 8) & 0xFF);
    b[r][2] = px_to_float<PROFILE>((v.x >> 16) & 0xFF);
    b[r][3] = px_to_float<PROFILE>(v.x >> 24);
    b[r][4] = px_to_float<PROFILE>(v.y & 0xFF);
    b[r][5] = px_to_float<PROFILE>((v.y >> 8) & 0xFF);
    b[r][6] = px_to_float<PROFILE>((v.y >> 16) & 0xFF);
    b[r][7] = px_to_float<PROFILE>(v.y >> 24);
  }

  constexpr int K = PROFILE == MDCT_PROFILE_REF_AVX ? K_AVX : (PROFILE == MDCT_PROFILE_REF_SSE ? K_SSE : K_TRUE);
  // STEREO tiers transpose first (T, rows, T, rows == columns then rows, :961-1004, :225-241);
  // Q32 and the encq tiers run rows then columns (:2158/:2189, :347-358, :1608-1636).
  if constexpr (LAYOUT == MDCT_LAYOUT_STEREO)
  {
    pass_cols<K>(b);
    pass_rows<K>(b);
  }
  else
  {
    pass_rows<K>(b);
    pass_cols<K>(b);
  }

  // Stored index s: natural v*8+u for Q32/STEREO, transposed u*8+v for the encq tiers,
  // with the quantiser taken AT THE STORED INDEX (:362, :1651).
#pragma unroll
  for (int s = 0; s < 64; s++)
  {
    const int hi = s >> 3, lo = s & 7;
    const float f = (LAYOUT == MDCT_LAYOUT_BLOCK || LAYOUT == MDCT_LAYOUT_BLOCK_SSE) ? b[lo][hi] : b[hi][lo];
    if constexpr (PROFILE == MDCT_PROFILE_REF_AVX)
      out[s] = quant_avx<SAFE>(f, qt.q[s]);
    else if constexpr (PROFILE == MDCT_PROFILE_REF_SSE)
      out[s] = quant_sse<SAFE>(f, qt.q[s]);
    else
      out[s] = quant_scalar(f, qt.q[s]);
  }
}

constexpr int kWG = 256;           // 4 waves
constexpr int kQ32RowStride = 72;  // 64 lanes + 8 pad: keeps rows 8-byte aligned for ds_read_b64

template <int PROFILE, int LAYOUT, bool SAFE>
__global__ __launch_bounds__(kWG) void k_fwd_quant_u8(U8Args a)
{
  const uint32_t t = blockIdx.x * kWG + threadIdx.x; // linear block index within the launch
  const bool valid = t < a.nblocks;
  const uint32_t lane = threadIdx.x & 63;

  // block coordinates.  STEREO enumerates (block row, eye, block x): simd_dct.cpp:1089-1099.
  uint32_t by, bx, eye = 0;
  {
    const uint32_t row = t / a.bpr;
    bx = t - row * a.bpr;
    if constexpr (LAYOUT == MDCT_LAYOUT_STEREO)
    {
      by = a.by0 + (row >> 1);
      eye = row & 1;
    }
    else
      by = a.by0 + row;
  }

  uint32_t q[64];
  if (valid)
  {
    const uint8_t *src = a.from + (size_t)by * 8 * a.pitch + (size_t)bx * 8;
    if constexpr (LAYOUT == MDCT_LAYOUT_STEREO)
      src += (size_t)eye * a.eye_offset;
    encode_block<PROFILE, LAYOUT, SAFE>(src, a.pitch, a.aligned8 != 0, a.qt, q);
  }

  if constexpr (LAYOUT == MDCT_LAYOUT_Q32)
  {
    // Output of 8 consecutive blocks (one reference "group") is 512 contiguous bytes
    // [coef*8 + blk] (:2227-2230) and group G of the plane sits at G*512, so a wave's
    // 64 blocks cover 4096 contiguous output bytes.  Stage them through wave-private
    // LDS as rows [coef][lane] so that every lane then stores 16 contiguous bytes.
    __shared__ __attribute__((aligned(16))) uint8_t lds[kWG / 64][64 * kQ32RowStride];
    uint8_t *wl = lds[threadIdx.x >> 6];
    if (valid)
    {
#pragma unroll
      for (int c = 0; c < 64; c++)
        wl[c * kQ32RowStride + lane] = (uint8_t)q[c];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    const uint32_t wave_t0 = t - lane;                                   // first block of this wave
    const uint32_t wave_blocks = wave_t0 < a.nblocks ? min(64u, a.nblocks - wave_t0) : 0u;
    const uint32_t wave_groups = wave_blocks >> 3;                        // nblocks % 8 == 0
    uint8_t *outw = a.to + ((size_t)a.by0 * a.bpr + wave_t0) * 64;         // == first group * 512
    const uint32_t c2 = (lane & 31) * 2;
#pragma unroll
    for (int k = 0; k < 4; k++)
    {
      const uint32_t g = 2 * k + (lane >> 5);
      if (g < wave_groups)
      {
        const uint2 lo = *reinterpret_cast<const uint2 *>(wl + c2 * kQ32RowStride + g * 8);
        const uint2 hi = *reinterpret_cast<const uint2 *>(wl + (c2 + 1) * kQ32RowStride + g * 8);
        *reinterpret_cast<uint4 *>(outw + g * 512 + c2 * 8) = make_uint4(lo.x, lo.y, hi.x, hi.y);
      }
    }
  }
  else if constexpr (LAYOUT == MDCT_LAYOUT_STEREO)
  {
    if (valid)
    {
      const size_t pos = ((size_t)by * 2 + eye) * a.bpr + bx;
#pragma unroll
      for (int i = 0; i < 64; i++)
        a.to[a.plane_stride * i + pos] = (uint8_t)q[i];
    }
  }
  else if constexpr (LAYOUT == MDCT_LAYOUT_BLOCK)
  {
    if (valid)
    {
      uint8_t *dst = a.to + (size_t)by * 8 * a.sizeX + (size_t)bx * 64;
#pragma unroll
      for (int k = 0; k < 4; k++)
      {
        uint32_t w[4];
#pragma unroll
        for (int j = 0; j < 4; j++)
          w[j] = q[k * 16 + j * 4] | (q[k * 16 + j * 4 + 1] << 8) | (q[k * 16 + j * 4 + 2] << 16) | (q[k * 16 + j * 4 + 3] << 24);
        *reinterpret_cast<uint4 *>(dst + k * 16) = make_uint4(w[0], w[1], w[2], w[3]);
      }
    }
  }
  else
  { // MDCT_LAYOUT_BLOCK_SSE (:1662-1676)
    if (valid)
    {
      const uint32_t ab = bx & 1;
      uint8_t *base = a.to + (size_t)by * 8 * a.sizeX + (size_t)(bx >> 1) * 128 + ab * 4;
      const bool spill = (by == a.by_last) && ((bx | 1u) == a.bpr - 1) && a.spill_ok;
#pragma unroll
      for (int i8 = 0; i8 < 8; i8++)
      {
        const uint32_t lo = q[i8 * 8 + 0] | (q[i8 * 8 + 1] << 8) | (q[i8 * 8 + 4] << 16) | (q[i8 * 8 + 5] << 24);
        *reinterpret_cast<uint32_t *>(base + i8 * 8) = lo;
        if (spill)
        {
          const uint32_t hi = q[i8 * 8 + 2] | (q[i8 * 8 + 3] << 8) | (q[i8 * 8 + 6] << 16) | (q[i8 * 8 + 7] << 24);
          *reinterpret_cast<uint32_t *>(base + 128 + i8 * 8) = hi;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// int16 / float32 engine-own kernels: plane layout in and out, 16 B per lane per row, so
// every global access is a fully coalesced 1 KiB wave transaction and no LDS is needed.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ int32_t sat_i16_rne(float v)
{
  return min(max(cvt_rne(v), -32768), 32767);
}

__device__ __forceinline__ void unpack_i16x8(const uint4 v, float (&row)[8])
{
  row[0] = (float)(int16_t)(v.x & 0xFFFF); row[1] = (float)(int16_t)(v.x >> 16);
  row[2] = (float)(int16_t)(v.y & 0xFFFF); row[3] = (float)(int16_t)(v.y >> 16);
  row[4] = (float)(int16_t)(v.z & 0xFFFF); row[5] = (float)(int16_t)(v.z >> 16);
  row[6] = (float)(int16_t)(v.w & 0xFFFF); row[7] = (float)(int16_t)(v.w >> 16);
}

__device__ __forceinline__ uint32_t pack2(int32_t lo, int32_t hi) { return ((uint32_t)lo & 0xFFFFu) | ((uint32_t)hi << 16); }

template <int MODE, bool HAS_LUT>
__device__ __forceinline__ void i16_block(const int16_t *src, int16_t *dst, size_t pitch_in, size_t pitch_out, const LutPair &lp)
{
  float b[8][8];
#pragma unroll
  for (int r = 0; r < 8; r++)
    unpack_i16x8(*reinterpret_cast<const uint4 *>(src + (size_t)r * pitch_in), b[r]);

  if constexpr (MODE == MODE_INV)
  {
    if constexpr (HAS_LUT)
    {
#pragma unroll
      for (int i = 0; i < 64; i++)
        b[i >> 3][i & 7] = b[i >> 3][i & 7] * lp.lut[i];
    }
  }
  else
  {
    pass_rows<K_OWN>(b);
    pass_cols<K_OWN>(b);
  }

  if constexpr (MODE == MODE_ROUNDTRIP && HAS_LUT)
  {
#pragma unroll
    for (int i = 0; i < 64; i++)
      b[i >> 3][i & 7] = (float)sat_i16_rne(b[i >> 3][i & 7] * lp.rq[i]) * lp.lut[i];
  }

  if constexpr (MODE != MODE_FWD)
  {
    ipass_cols(b);
    ipass_rows(b);
  }

#pragma unroll
  for (int r = 0; r < 8; r++)
  {
    int32_t o[8];
#pragma unroll
    for (int c = 0; c < 8; c++)
    {
      float f = b[r][c];
      if constexpr (MODE == MODE_FWD && HAS_LUT)
        f = f * lp.rq[r * 8 + c];
      o[c] = sat_i16_rne(f);
    }
    *reinterpret_cast<uint4 *>(dst + (size_t)r * pitch_out) = make_uint4(pack2(o[0], o[1]), pack2(o[2], o[3]), pack2(o[4], o[5]), pack2(o[6], o[7]));
  }
}

template <int MODE, bool HAS_LUT>
__global__ __launch_bounds__(kWG) void k_i16(I16Args a)
{
  const uint32_t t = blockIdx.x * kWG + threadIdx.x;
  if (t >= a.nblocks)
    return;
  const uint32_t row = t / a.bpr;
  const uint32_t bx = t - row * a.bpr;
  const size_t by = a.by0 + row;
  i16_block<MODE, HAS_LUT>(a.from + by * 8 * a.pitch_in + (size_t)bx * 8, a.to + by * 8 * a.pitch_out + (size_t)bx * 8, a.pitch_in, a.pitch_out, a.lp);
}

// Several planes (each with its own table) in one launch: linear block index over the
// concatenation of the planes; prefix[] is the exclusive scan of per-plane block counts.
__global__ __launch_bounds__(kWG) void k_i16_planes(PlaneBatchArgs a)
{
  const uint32_t t = blockIdx.x * kWG + threadIdx.x;
  // plane index from the wave's first block: wave-uniform, so the table reads stay scalar
  const uint32_t tw = __builtin_amdgcn_readfirstlane(t - (threadIdx.x & 63));
  if (tw >= a.prefix[a.n])
    return;
  int p = 0;
#pragma unroll
  for (int i = 1; i < kMaxPlanes; i++)
    p += (i < a.n && tw >= a.prefix[i]) ? 1 : 0;
  const uint32_t lt = t - a.prefix[p];
  if (lt >= a.nblk[p])
    return;
  const uint32_t bpr = a.bpr[p];
  const uint32_t row = lt / bpr;
  const uint32_t bx = lt - row * bpr;
  const size_t pin = a.pitch_in[p], pout = a.pitch_out[p];
  // p is wave-uniform (planes are padded to whole waves): scalar table reads, scalar branch
  const int16_t *src = a.from[p] + (size_t)row * 8 * pin + (size_t)bx * 8;
  int16_t *dst = a.to[p] + (size_t)row * 8 * pout + (size_t)bx * 8;
  if (a.has_lut[p])
    i16_block<MODE_ROUNDTRIP, true>(src, dst, pin, pout, a.lp[p]);
  else
    i16_block<MODE_ROUNDTRIP, false>(src, dst, pin, pout, a.lp[p]);
}

template <int MODE>
__global__ __launch_bounds__(kWG) void k_f32(F32Args a)
{
  const uint32_t t = blockIdx.x * kWG + threadIdx.x;
  if (t >= a.nblocks)
    return;
  const uint32_t row = t / a.bpr;
  const uint32_t bx = t - row * a.bpr;
  const size_t by = a.by0 + row;
  const float *src = a.from + by * 8 * a.pitch_in + (size_t)bx * 8;
  float *dst = a.to + by * 8 * a.pitch_out + (size_t)bx * 8;

  float b[8][8];
#pragma unroll
  for (int r = 0; r < 8; r++)
  {
    const float4 lo = *reinterpret_cast<const float4 *>(src + (size_t)r * a.pitch_in);
    const float4 hi = *reinterpret_cast<const float4 *>(src + (size_t)r * a.pitch_in + 4);
    b[r][0] = lo.x; b[r][1] = lo.y; b[r][2] = lo.z; b[r][3] = lo.w;
    b[r][4] = hi.x; b[r][5] = hi.y; b[r][6] = hi.z; b[r][7] = hi.w;
  }
  if constexpr (MODE == MODE_FWD)
  {
    pass_rows<K_OWN>(b);
    pass_cols<K_OWN>(b);
  }
  else
  {
    ipass_cols(b);
    ipass_rows(b);
  }
#pragma unroll
  for (int r = 0; r < 8; r++)
  {
    *reinterpret_cast<float4 *>(dst + (size_t)r * a.pitch_out) = make_float4(b[r][0], b[r][1], b[r][2], b[r][3]);
    *reinterpret_cast<float4 *>(dst + (size_t)r * a.pitch_out + 4) = make_float4(b[r][4], b[r][5], b[r][6], b[r][7]);
  }
}

// read-N / write-N stream copy, 16 B per lane, grid-stride: the box's measured HBM roofline.
__global__ __launch_bounds__(kWG) void k_stream_copy(const uint4 *__restrict__ from, uint4 *__restrict__ to, size_t n16)
{
  const size_t stride = (size_t)gridDim.x * kWG;
  for (size_t i = (size_t)blockIdx.x * kWG + threadIdx.x; i < n16; i += stride)
    to[i] = from[i];
}

// ---------------------------------------------------------------------------------------
// launchers (host)
// ---------------------------------------------------------------------------------------
static inline uint32_t grid_for(uint32_t nblocks) { return (nblocks + kWG - 1) / kWG; }

template <int PROFILE, int LAYOUT>
static hipError_t launch_u8_pl(const U8Args &a, bool safe, hipStream_t s)
{
  if (safe)
    hipLaunchKernelGGL((k_fwd_quant_u8<PROFILE, LAYOUT, true>), dim3(grid_for(a.nblocks)), dim3(kWG), 0, s, a);
  else
    hipLaunchKernelGGL((k_fwd_quant_u8<PROFILE, LAYOUT, false>), dim3(grid_for(a.nblocks)), dim3(kWG), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_fwd_quant_u8(const U8Args &a, int layout, int profile, bool safe, hipStream_t s)
{
  if (a.nblocks == 0)
    return hipSuccess;
  if (layout == MDCT_LAYOUT_Q32 && profile == MDCT_PROFILE_REF_AVX)
    return launch_u8_pl<MDCT_PROFILE_REF_AVX, MDCT_LAYOUT_Q32>(a, safe, s);
  if (layout == MDCT_LAYOUT_STEREO && profile == MDCT_PROFILE_REF_SSE)
    return launch_u8_pl<MDCT_PROFILE_REF_SSE, MDCT_LAYOUT_STEREO>(a, safe, s);
  if (layout == MDCT_LAYOUT_STEREO && profile == MDCT_PROFILE_REF_SCALAR)
    return launch_u8_pl<MDCT_PROFILE_REF_SCALAR, MDCT_LAYOUT_STEREO>(a, false, s);
  if (layout == MDCT_LAYOUT_BLOCK && profile == MDCT_PROFILE_REF_SCALAR)
    return launch_u8_pl<MDCT_PROFILE_REF_SCALAR, MDCT_LAYOUT_BLOCK>(a, false, s);
  if (layout == MDCT_LAYOUT_BLOCK_SSE && profile == MDCT_PROFILE_REF_SSE)
    return launch_u8_pl<MDCT_PROFILE_REF_SSE, MDCT_LAYOUT_BLOCK_SSE>(a, safe, s);
  return hipErrorInvalidValue;
}

template <int MODE>
static hipError_t launch_i16_m(const I16Args &a, bool has_lut, hipStream_t s)
{
  if (has_lut)
    hipLaunchKernelGGL((k_i16<MODE, true>), dim3(grid_for(a.nblocks)), dim3(kWG), 0, s, a);
  else
    hipLaunchKernelGGL((k_i16<MODE, false>), dim3(grid_for(a.nblocks)), dim3(kWG), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_i16(const I16Args &a, int mode, bool has_lut, hipStream_t s)
{
  if (a.nblocks == 0)
    return hipSuccess;
  switch (mode)
  {
  case MODE_FWD: return launch_i16_m<MODE_FWD>(a, has_lut, s);
  case MODE_INV: return launch_i16_m<MODE_INV>(a, has_lut, s);
  case MODE_ROUNDTRIP: return launch_i16_m<MODE_ROUNDTRIP>(a, has_lut, s);
  }
  return hipErrorInvalidValue;
}

hipError_t launch_i16_planes(const PlaneBatchArgs &a, hipStream_t s)
{
  const uint32_t total = a.prefix[a.n];
  if (total == 0)
    return hipSuccess;
  hipLaunchKernelGGL(k_i16_planes, dim3(grid_for(total)), dim3(kWG), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_f32(const F32Args &a, int mode, hipStream_t s)
{
  if (a.nblocks == 0)
    return hipSuccess;
  if (mode == MODE_FWD)
    hipLaunchKernelGGL((k_f32<MODE_FWD>), dim3(grid_for(a.nblocks)), dim3(kWG), 0, s, a);
  else
    hipLaunchKernelGGL((k_f32<MODE_INV>), dim3(grid_for(a.nblocks)), dim3(kWG), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_stream_copy(const void *from, void *to, size_t bytes, int cus, hipStream_t s)
{
  const size_t n16 = bytes / 16;
  if (n16 == 0)
    return hipSuccess;
  size_t grid = (n16 + kWG - 1) / kWG;
  const size_t cap = (size_t)cus * 8;
  if (grid > cap)
    grid = cap;
  hipLaunchKernelGGL(k_stream_copy, dim3((uint32_t)grid), dim3(kWG), 0, s, (const uint4 *)from, (uint4 *)to, n16);
  return hipGetLastError();
}

} // namespace mdct
