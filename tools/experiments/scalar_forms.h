// scalar_forms.h -- the u8 tiers written one float operation per line (the form the product used in round 1).
// NOT part of the product: the kernels in simd_dct_amd/csrc/mdct_kernels.hip compute the same values with
// packed fp32 (dct8_h / dct8_v).  Kept for the A/B harnesses under tools/ (compute-only passes, variants),
// which #include the product kernels first and this file second.
#pragma once

namespace mdct
{

// ---------------------------------------------------------------------------------------
// 1-D 8-point forward kernel on eight registers.
// ---------------------------------------------------------------------------------------
template <int K>
__device__ __forceinline__ void dct8(const DctConsts &C, float &p0, float &p1, float &p2, float &p3, float &p4, float &p5, float &p6, float &p7)
{
  const float kCa = C.a, kCb = C.b, kCc = C.c, kCd = C.d, kCe = C.e, kCf = C.f, kCn = C.n;
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
    else
    {
      static_assert(K == K_AVX, "unknown 1-D kernel");
      o1 = t1 + ((kCd * x25m) - (kCf * x43m));
      o3 = t3 - ((kCa * x25m) + (kCd * x43m)); // quirk: -Cd
    }
    o5 = t5 + ((kCf * x25m) - (kCc * x43m));
    o7 = t7 + ((kCc * x25m) + (kCa * x43m));
  }

  p0 = kCn * o0; p1 = kCn * o1; p2 = kCn * o2; p3 = kCn * o3;
  p4 = kCn * o4; p5 = kCn * o5; p6 = kCn * o6; p7 = kCn * o7;
}


template <int K>
__device__ __forceinline__ void pass_rows(const DctConsts &C, float (&b)[8][8])
{
#pragma unroll
  for (int r = 0; r < 8; r++)
    dct8<K>(C, b[r][0], b[r][1], b[r][2], b[r][3], b[r][4], b[r][5], b[r][6], b[r][7]);
}

template <int K>
__device__ __forceinline__ void pass_cols(const DctConsts &C, float (&b)[8][8])
{
#pragma unroll
  for (int c = 0; c < 8; c++)
    dct8<K>(C, b[0][c], b[1][c], b[2][c], b[3][c], b[4][c], b[5][c], b[6][c], b[7][c]);
}


// B1 :2224  clamp(127 + rne(f*q), 0, 255)   (wrapping int32 add, like _mm256_add_epi32)
template <bool SAFE>
__device__ __forceinline__ uint32_t quant_avx(float f, float q, float magic23)
{
  const float v = f * q;
  if constexpr (SAFE)
    return (uint32_t)clamp255((int32_t)((uint32_t)cvtps_epi32_exact(v) + 127u));
  else
    return __float_as_uint(__builtin_amdgcn_fmed3f(v, -127.0f, 128.0f) + magic23) + 127u;
}

// B2/B3 :1020  clamp(rne(f*q + 127.0f), 0, 255)
template <bool SAFE>
__device__ __forceinline__ uint32_t quant_sse(float f, float q, float magic23)
{
  const float v = (f * q) + 127.0f;
  if constexpr (SAFE)
    return (uint32_t)clamp255(cvtps_epi32_exact(v));
  else
    return __float_as_uint(__builtin_amdgcn_fmed3f(v, 0.0f, 255.0f) + magic23);
}

// B4/B5 :245, :362  (uint8_t)roundf(_clamp(f*qs + 127/255, 0, 1) * 255)
// roundf (half away from zero) of x in [0, 255] without libm: r = rne(x) by the magic add, and
// the two differ only at an exact tie that rne resolved downwards (x - r == +0.5), where roundf
// wants r + 1.  x - r is exact (|x - r| <= 0.5 and both are multiples of ulp(x)).
__device__ __forceinline__ uint32_t quant_scalar(float f, float qs, float magic23)
{
  float v = (f * qs) + (127.0f / 255.0f);
  v = v > 0.f ? (v < 1.f ? v : 1.f) : 0.f; // _clamp(v, 0, 1) of :50-54, NaN -> 0
  const float x = v * 255.f;
  const float t = x + magic23;
  const float r = t - magic23;
  return __float_as_uint(t) + ((x - r) == 0.5f ? 1u : 0u);
}


template <int PROFILE>
__device__ __forceinline__ float px_to_float(float f)
{
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

// convert (consumes `rows`), both passes in the profile's order, quantise: out[s] = word whose low
// byte is the coefficient at stored index s
template <int PROFILE, int LAYOUT, bool SAFE>
__device__ __forceinline__ void encode_rows(const DctConsts &C, const uint2 (&rows)[8], const QuantTable &qt, const float *px_div255, float (&b)[8][8])
{
#pragma unroll
  for (int r = 0; r < 8; r++)
  {
    const uint2 v = rows[r];
    if constexpr (PROFILE == MDCT_PROFILE_REF_SCALAR)
    { // px / 255.f (:222, :343) has only 256 possible results: looked up, not divided (see kernel)
      b[r][0] = px_div255[v.x & 0xFF];
      b[r][1] = px_div255[(v.x >> 8) & 0xFF];
      b[r][2] = px_div255[(v.x >> 16) & 0xFF];
      b[r][3] = px_div255[v.x >> 24];
      b[r][4] = px_div255[v.y & 0xFF];
      b[r][5] = px_div255[(v.y >> 8) & 0xFF];
      b[r][6] = px_div255[(v.y >> 16) & 0xFF];
      b[r][7] = px_div255[v.y >> 24];
    }
    else
    {
      b[r][0] = px_to_float<PROFILE>(ubyte_to_float<0>(v.x));
      b[r][1] = px_to_float<PROFILE>(ubyte_to_float<1>(v.x));
      b[r][2] = px_to_float<PROFILE>(ubyte_to_float<2>(v.x));
      b[r][3] = px_to_float<PROFILE>(ubyte_to_float<3>(v.x));
      b[r][4] = px_to_float<PROFILE>(ubyte_to_float<0>(v.y));
      b[r][5] = px_to_float<PROFILE>(ubyte_to_float<1>(v.y));
      b[r][6] = px_to_float<PROFILE>(ubyte_to_float<2>(v.y));
      b[r][7] = px_to_float<PROFILE>(ubyte_to_float<3>(v.y));
    }
  }

  (void)C; (void)qt;
}

template <int PROFILE, int LAYOUT, bool SAFE>
__device__ __forceinline__ void transform_quantise(const DctConsts &C, float (&b)[8][8], const QuantTable &qt, uint32_t (&out)[64])
{
  constexpr int K = PROFILE == MDCT_PROFILE_REF_AVX ? K_AVX : (PROFILE == MDCT_PROFILE_REF_SSE ? K_SSE : K_TRUE);
  // STEREO tiers transpose first (T, rows, T, rows == columns then rows, :961-1004, :225-241);
  // Q32 and the encq tiers run rows then columns (:2158/:2189, :347-358, :1608-1636).
  if constexpr (LAYOUT == MDCT_LAYOUT_STEREO)
  {
    pass_cols<K>(C, b);
    pass_rows<K>(C, b);
  }
  else
  {
    pass_rows<K>(C, b);
    pass_cols<K>(C, b);
  }

  // Stored index s: natural v*8+u for Q32/STEREO, transposed u*8+v for the encq tiers,
  // with the quantiser taken AT THE STORED INDEX (:362, :1651).
#pragma unroll
  for (int s = 0; s < 64; s++)
  {
    const int hi = s >> 3, lo = s & 7;
    const float f = (LAYOUT == MDCT_LAYOUT_BLOCK || LAYOUT == MDCT_LAYOUT_BLOCK_SSE) ? b[lo][hi] : b[hi][lo];
    if constexpr (PROFILE == MDCT_PROFILE_REF_AVX)
      out[s] = quant_avx<SAFE>(f, qt.q[s], C.magic23);
    else if constexpr (PROFILE == MDCT_PROFILE_REF_SSE)
      out[s] = quant_sse<SAFE>(f, qt.q[s], C.magic23);
    else
      out[s] = quant_scalar(f, qt.q[s], C.magic23);
  }
}

template <int PROFILE, int LAYOUT, bool SAFE>
__device__ __forceinline__ void encode_block(const DctConsts &C, const uint8_t *src, size_t pitch, const QuantTable &qt, const float *px_div255, uint32_t (&out)[64])
{
  uint2 rows[8];
  load_block_rows(src, pitch, rows);
  float b[8][8];
  encode_rows<PROFILE, LAYOUT, SAFE>(C, rows, qt, px_div255, b);
  transform_quantise<PROFILE, LAYOUT, SAFE>(C, b, qt, out);
}


} // namespace mdct
