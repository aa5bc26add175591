/*
 * dct_oracle.c -- CPU restatement of the reference's 8x8 block-DCT hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see dct_oracle.h).  Plain C, scalar, one rounding per
 * written float operation: build with `-O2 -ffp-contract=off`, never -ffast-math.
 * Each function cites the reference lines (under /root/reference/src/) it restates.
 * It was written from the behaviour of those lines, not translated from them: the
 * reference is SIMD-intrinsic code, this is a per-block scalar description.
 */
#include "dct_oracle.h"

#include <math.h>
#include <string.h>

/* simd_dct.cpp:140-146 (same literals in every tier: :2076-2082, :436-442). */
static const float Ca = 1.3870398453221474618216191915664f;  /* sqrt2*cos(1*pi/16) */
static const float Cb = 1.3065629648763765278566431734272f;  /* sqrt2*cos(2*pi/16) */
static const float Cc = 1.1758756024193587169744671046113f;  /* sqrt2*cos(3*pi/16) */
static const float Cd = 0.78569495838710218127789736765722f; /* sqrt2*cos(5*pi/16) */
static const float Ce = 0.54119610014619698439972320536639f; /* sqrt2*cos(6*pi/16) */
static const float Cf = 0.27589937928294301233595756366937f; /* sqrt2*cos(7*pi/16) */
static const float Cn = 0.35355339059327376220042218105242f; /* 1/sqrt(8) */

enum { K_AVX = 0, K_SSE = 1, K_TRUE = 2, K_OWN = 3 };


/* ------------------------------------------------------------------------------------------
 * Engine-own 1-D kernels [unpinned]: the Arai-Agui-Nakajima scaled DCT (5 multiplies and 29
 * additions per 8 points, as popularised by the IJG float DCT).  VALU issue -- not HBM -- bounds
 * the 8-bit kernels that use it on gfx950 and a packed fused multiply-add issues in the time of a
 * packed multiply, so from round 6 on every multiply whose product feeds one or two additions is
 * FUSED with them: 4 of the 5 multiplies per pass ride an fma (30 instead of 34 operations; the
 * fifth, z5, feeds two fmas as the addend).  Every written operation -- add, sub, mul, fma -- is
 * rounded once; fma(a, b, c) is the IEEE fusedMultiplyAdd (C99 fmaf), fma(-a, b, c) negates exactly.
 * The engine's kernels (csrc/mdct_kernels.hip: aan_fwd8 / aan_inv8 and their packed forms) perform
 * the same operations on the same operands.
 *   forward:  y_k = sqrt(8) * a_k * X_k      (X = orthonormal DCT-II, a_0 = 1, a_k = sqrt2*cos(k*pi/16))
 *   inverse:  takes z_k = a_k * X_k / sqrt(8) ... per dimension; the 2-D tables below carry
 *             the factors so that forward-table * inverse-table == 1/64 exactly.
 * ---------------------------------------------------------------------------------------- */
const double orc_aan_scale[8] = {1.0, 1.387039845322148, 1.306562964876377, 1.175875602419359,
                                 1.0, 0.785694958387102, 0.541196100146197, 0.275899379282943};

static const float A_707 = 0.707106781186547524f;  /* cos(pi/4)            */
static const float A_382 = 0.382683432365089772f;  /* cos(3pi/8)           */
static const float A_541 = 0.541196100146196985f;  /* cos(pi/8)-cos(3pi/8) */
static const float A_1306 = 1.306562964876376528f; /* cos(pi/8)+cos(3pi/8) */
static const float A_1414 = 1.414213562373095049f; /* sqrt(2)              */
static const float A_1847 = 1.847759065022573512f; /* 2*cos(pi/8)          */
static const float A_1082 = 1.082392200292393968f; /* 2*(cos(pi/8)-cos(3pi/8)) */
static const float A_2613 = 2.613125929752753056f; /* 2*(cos(pi/8)+cos(3pi/8)) */

/* The functions that reach the butterflies are compiled twice (GCC function multi-versioning): for hosts with FMA3, where
 * fmaf is one instruction, and for any x86-64, where it is libm's exactly rounded software fmaf.  Same bits either way. */
#if defined(__x86_64__) && defined(__GNUC__) && !defined(__clang__) && !defined(ORC_NO_CLONES)
#define ORC_FMA_CLONES __attribute__((target_clones("fma", "default")))
#else
#define ORC_FMA_CLONES
#endif

static inline __attribute__((always_inline)) void aan_fwd8(float *p, ptrdiff_t s)
{
  const float d0 = p[0], d1 = p[s], d2 = p[2 * s], d3 = p[3 * s], d4 = p[4 * s], d5 = p[5 * s], d6 = p[6 * s], d7 = p[7 * s];
  const float t0 = d0 + d7, t7 = d0 - d7, t1 = d1 + d6, t6 = d1 - d6;
  const float t2 = d2 + d5, t5 = d2 - d5, t3 = d3 + d4, t4 = d3 - d4;
  /* even part */
  const float e10 = t0 + t3, e13 = t0 - t3, e11 = t1 + t2, e12 = t1 - t2;
  const float s1 = e12 + e13;
  /* odd part */
  const float o10 = t4 + t5, o11 = t5 + t6, o12 = t6 + t7;
  const float z5 = (o10 - o12) * A_382;
  const float z2 = fmaf(A_541, o10, z5);
  const float z4 = fmaf(A_1306, o12, z5);
  const float z11 = fmaf(o11, A_707, t7), z13 = fmaf(-o11, A_707, t7);
  p[0] = e10 + e11;
  p[4 * s] = e10 - e11;
  p[2 * s] = fmaf(s1, A_707, e13);
  p[6 * s] = fmaf(-s1, A_707, e13);
  p[5 * s] = z13 + z2;
  p[3 * s] = z13 - z2;
  p[s] = z11 + z4;
  p[7 * s] = z11 - z4;
}

static inline __attribute__((always_inline)) void aan_inv8(float *p, ptrdiff_t s)
{
  const float i0 = p[0], i1 = p[s], i2 = p[2 * s], i3 = p[3 * s], i4 = p[4 * s], i5 = p[5 * s], i6 = p[6 * s], i7 = p[7 * s];
  /* even part */
  const float e10 = i0 + i4, e11 = i0 - i4;
  const float e13 = i2 + i6;
  const float e12 = fmaf(i2 - i6, A_1414, -e13);
  const float t0 = e10 + e13, t3 = e10 - e13, t1 = e11 + e12, t2 = e11 - e12;
  /* odd part */
  const float z13 = i5 + i3, z10 = i5 - i3, z11 = i1 + i7, z12 = i1 - i7;
  const float t7 = z11 + z13;
  const float z5 = (z10 + z12) * A_1847;
  const float o10 = fmaf(A_1082, z12, -z5);
  const float o12 = fmaf(-A_2613, z10, z5);
  const float t6 = o12 - t7;
  const float t5 = fmaf(z11 - z13, A_1414, -t6);
  const float t4 = o10 + t5;
  p[0] = t0 + t7;
  p[7 * s] = t0 - t7;
  p[s] = t1 + t6;
  p[6 * s] = t1 - t6;
  p[2 * s] = t2 + t5;
  p[5 * s] = t2 - t5;
  p[4 * s] = t3 + t4;
  p[3 * s] = t3 - t4;
}

ORC_FMA_CLONES void orc_aan_fwd8(float *p, ptrdiff_t s) { aan_fwd8(p, s); }
ORC_FMA_CLONES void orc_aan_inv8(float *p, ptrdiff_t s) { aan_inv8(p, s); }

/* 2-D tables, index v*8+u.  fwd: raw AAN output -> orthonormal coefficient; inv: orthonormal
 * coefficient -> AAN inverse input (includes the 1/8 of the two inverse passes).  Products of
 * doubles rounded once to float, so host (product) and oracle agree bit for bit. */
void orc_aan_tables(float *fwd, float *inv)
{
  for (int v = 0; v < 8; v++)
    for (int u = 0; u < 8; u++)
    {
      const double a = orc_aan_scale[v] * orc_aan_scale[u];
      fwd[v * 8 + u] = (float)(1.0 / (8.0 * a));
      inv[v * 8 + u] = (float)(a / 8.0);
    }
}

void orc_dct8(float *p, ptrdiff_t s, int which)
{
  if (which == K_OWN)
  { /* engine-own: scaled AAN butterfly followed by the 1-D scale factors */
    orc_aan_fwd8(p, s);
    for (int k = 0; k < 8; k++)
      p[k * s] = p[k * s] * (float)(1.0 / (2.8284271247461900976 * orc_aan_scale[k]));
    return;
  }
  const float p0 = p[0], p1 = p[s], p2 = p[2 * s], p3 = p[3 * s];
  const float p4 = p[4 * s], p5 = p[5 * s], p6 = p[6 * s], p7 = p[7 * s];

  /* first two butterfly stages are identical in all tiers
   * (simd_dct.cpp:148-161, :2160-2173; the SSE tier builds x61m / x43m as
   * (-p1)+p6 and (-p3)+p4 at :470-484, which is the same IEEE value). */
  const float x07p = p0 + p7, x16p = p1 + p6, x25p = p2 + p5, x34p = p3 + p4;
  const float x07m = p0 - p7, x61m = p6 - p1, x25m = p2 - p5, x43m = p4 - p3;
  const float pp = x07p + x34p, pm = x07p - x34p;
  const float qp = x16p + x25p, qm = x16p - x25p;

  float o0, o1, o2, o3, o4, o5, o6, o7;

  o0 = pp + qp;
  o4 = pp - qp;

  switch (which)
  {
  case K_TRUE: /* simd_dct.cpp:163-171, C left-to-right association */
    o2 = Cb * pm + Ce * qm;
    o6 = Ce * pm - Cb * qm;
    o1 = ((Ca * x07m - Cc * x61m) + Cd * x25m) - Cf * x43m;
    o3 = ((Cc * x07m + Cf * x61m) - Ca * x25m) + Cd * x43m;
    o5 = ((Cd * x07m + Ca * x61m) + Cf * x25m) - Cc * x43m;
    o7 = ((Cf * x07m + Cd * x61m) + Cc * x25m) + Ca * x43m;
    break;

  case K_SSE: /* simd_dct.cpp:547-577 (factor rows :547-550, pairwise sum :576-577,
               * even part :585-608).  Lane 0 of xf_7_factors is +C_f: k=1 sign quirk. */
    o2 = (Cb * pm) + (Ce * qm);
    o6 = (Ce * pm) + ((-Cb) * qm);
    o1 = ((Ca * x07m) + ((-Cc) * x61m)) + ((Cd * x25m) + (Cf * x43m));
    o3 = ((Cc * x07m) + (Cf * x61m)) + (((-Ca) * x25m) + (Cd * x43m));
    o5 = ((Cd * x07m) + (Ca * x61m)) + ((Cf * x25m) + ((-Cc) * x43m));
    o7 = ((Cf * x07m) + (Cd * x61m)) + ((Cc * x25m) + (Ca * x43m));
    break;

  default: /* K_AVX: simd_dct.cpp:2176-2183 (== :1972-1979, AVX-512VL).
            * o3 subtracts (Ca*x25m + Cd*x43m): k=3 sign quirk. */
    o2 = (Cb * pm) + (Ce * qm);
    o6 = (Ce * pm) - (Cb * qm);
    o1 = ((Ca * x07m) - (Cc * x61m)) + ((Cd * x25m) - (Cf * x43m));
    o3 = ((Cc * x07m) + (Cf * x61m)) - ((Ca * x25m) + (Cd * x43m));
    o5 = ((Cd * x07m) + (Ca * x61m)) + ((Cf * x25m) - (Cc * x43m));
    o7 = ((Cf * x07m) + (Cd * x61m)) + ((Cc * x25m) + (Ca * x43m));
    break;
  }

  p[0] = Cn * o0;     p[s] = Cn * o1;     p[2 * s] = Cn * o2; p[3 * s] = Cn * o3;
  p[4 * s] = Cn * o4; p[5 * s] = Cn * o5; p[6 * s] = Cn * o6; p[7 * s] = Cn * o7;
}

/* 1-D inverse of K_OWN (for the unit tests): orthonormal coefficients in, samples out. */
void orc_idct8_own(float *p, ptrdiff_t s)
{
  for (int k = 0; k < 8; k++)
    p[k * s] = p[k * s] * (float)(orc_aan_scale[k] / 2.8284271247461900976);
  orc_aan_inv8(p, s);
}

/* x86 cvtps_epi32 under default MXCSR: round-to-nearest-even; out of range or NaN
 * gives the "integer indefinite" 0x80000000 (simd_dct.cpp:2224, :1020). */
static int32_t cvtps_epi32(float v)
{
  if (!(fabsf(v) < 2147483648.0f))
    return INT32_MIN;
  return (int32_t)rintf(v);
}

static int32_t clamp_0_255(int32_t v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

static void rows(float *blk, int which) { for (int r = 0; r < 8; r++) orc_dct8(blk + r * 8, 1, which); }
static void cols(float *blk, int which) { for (int c = 0; c < 8; c++) orc_dct8(blk + c, 8, which); }
static void transpose(float *blk)
{
  for (int i = 0; i < 8; i++)
    for (int j = i + 1; j < 8; j++)
    {
      const float t = blk[i * 8 + j];
      blk[i * 8 + j] = blk[j * 8 + i];
      blk[j * 8 + i] = t;
    }
}

static int check_args(const void *from, const void *to, size_t sizeX, size_t sizeY, size_t xmul)
{
  if (from == NULL || to == NULL)
    return 1; /* sdr_InvalidParameter, simd_dct.cpp:75, :97, :117 */
  if (sizeX % xmul != 0 || sizeY % 8 != 0)
    return 2; /* sdr_NotSupported, simd_dct.cpp:76, :98, :118 */
  return 0;
}

/* ------------------------------------------------------------------ B1 ---- */
/* simd_dct.cpp:2064-2262.  (float)px, K_AVX rows then K_AVX columns, i = v*8+u,
 * q[i] = 255.0f/(lut[i]*0.95f) (:2239), byte = clamp(127 + rne(f*q[i]), 0, 255) (:2224),
 * stored at pTo[y*W + g*512 + i*8 + b] (:2227-2230). */
static void q32_block(const uint8_t *src, size_t pitch, const float *q, uint8_t *dst /* stride 8 */)
{
  float blk[64];
  for (int r = 0; r < 8; r++)
    for (int c = 0; c < 8; c++)
      blk[r * 8 + c] = (float)src[r * pitch + c];
  rows(blk, K_AVX);
  cols(blk, K_AVX);
  for (int i = 0; i < 64; i++)
    dst[i * 8] = (uint8_t)clamp_0_255(127 + cvtps_epi32(blk[i] * q[i]));
}

static void make_q255(const float *lut, float *q)
{
  for (int i = 0; i < 64; i++)
    q[i] = 255.0f / (lut[i] * 0.95f);
}

int orc_q32_avx(const uint8_t *from, uint8_t *to, const float *lut, size_t W, size_t H, size_t startY, size_t endY)
{
  const int e = check_args(from, to, W, H, 64);
  if (e)
    return e;
  float q[64];
  make_q255(lut, q);
  /* row loop :2243-2261: y < H/2, processed iff startY <= 2y <= endY */
  for (size_t y = 0; y < H / 2; y += 8)
  {
    if (y * 2 < startY)
      continue;
    if (y * 2 > endY)
      break;
    for (size_t g = 0; g < W / 64; g++)
      for (size_t b = 0; b < 8; b++)
        q32_block(from + y * W + g * 64 + b * 8, W, q, to + y * W + g * 512 + b);
  }
  return 0;
}

int orc_q32_native(const uint8_t *from, uint8_t *to, size_t pitch_in, const float *lut, size_t W, size_t H, size_t by0, size_t by1)
{
  const int e = check_args(from, to, W, H, 64);
  if (e)
    return e;
  if (by1 > H / 8 || by0 > by1 || pitch_in < W)
    return 1;
  float q[64];
  make_q255(lut, q);
  for (size_t by = by0; by < by1; by++)
    for (size_t g = 0; g < W / 64; g++)
      for (size_t b = 0; b < 8; b++)
        q32_block(from + by * 8 * pitch_in + g * 64 + b * 8, pitch_in, q, to + by * 8 * W + g * 512 + b);
  return 0;
}

/* ------------------------------------------------------------ B2 / B3 ----- */
/* shared SSE block arithmetic: px*(1/255) (:949-957), quantise
 * clamp(rne(f*q + 127.0f), 0, 255) with max(0,.) then min(255,.) on int32 (:1020). */
static uint8_t sse_quant(float f, float q) { return (uint8_t)clamp_0_255(cvtps_epi32((f * q) + 127.0f)); }

static void sse_load(const uint8_t *src, size_t pitch, float *blk)
{
  const float inv255 = 1.f / (float)0xFF;
  for (int r = 0; r < 8; r++)
    for (int c = 0; c < 8; c++)
      blk[r * 8 + c] = inv255 * (float)src[r * pitch + c];
}

/* simd_dct.cpp:896-1103 (SSE4.1), == :1106-1327 (SSE2), :1330-1536 (SSSE3).
 * T, K_SSE rows, T, K_SSE rows (:961-1004): i = v*8+u.  Output coefficient-planar:
 * plane i at to + (W*H/64)*i, position advances one byte per block in the order
 * (block row, eye, block x) (:1061-1099); skipped block rows still advance the
 * position by W/4 (:1075-1081). */
int orc_stereo_sse(const uint8_t *from, uint8_t *to, const float *lut, size_t W, size_t H, size_t startY, size_t endY)
{
  const int e = check_args(from, to, W, H, 8);
  if (e)
    return e;
  if (W % 16 != 0 || H % 16 != 0)
    return 2; /* the reference walks 16 px at a time (:945) and reads the second image from row
               * H/2 (:1097): it over-reads the plane otherwise; refused here and in the product */
  float q[64];
  make_q255(lut, q);
  const size_t plane = (W * H) / 64;
  size_t pos = 0;
  for (size_t y = 0; y < H / 2; y += 8)
  {
    if (y * 2 < startY)
    {
      pos += W / 4;
      continue;
    }
    if (y * 2 > endY)
      break;
    for (int eye = 0; eye < 2; eye++)
      for (size_t bx = 0; bx < W / 8; bx++)
      {
        float blk[64];
        sse_load(from + (size_t)eye * (W * H / 2) + y * W + bx * 8, W, blk);
        transpose(blk); rows(blk, K_SSE); transpose(blk); rows(blk, K_SSE);
        for (int i = 0; i < 64; i++)
          to[plane * i + pos] = sse_quant(blk[i], q[i]);
        pos++;
      }
  }
  return 0;
}

/* simd_dct.cpp:1540-1704 (SSE4.1) == :1707-1864 (SSSE3).  K_SSE rows, T, K_SSE rows:
 * result[i8][j] with i8 = u, j = v, quantised with q[i8*8+j] (:1651-1655).  The store
 * (:1662-1676) writes, per 16-px block pair (A,B) and coefficient row i8, the bytes
 * {A[i8][0],A[i8][1],A[i8][4],A[i8][5],B[..same..]} at pair*128 + i8*8, and columns
 * {2,3,6,7} 128 bytes further on, where the NEXT pair (or next line) overwrites them;
 * only the spill of the last pair of the last processed line survives. */
int orc_encq_sse(const uint8_t *from, uint8_t *to, const float *lut, size_t W, size_t H, size_t startY, size_t endY)
{
  const int e = check_args(from, to, W, H, 8);
  if (e)
    return e;
  if (W % 16 != 0)
    return 2;
  float q[64];
  make_q255(lut, q);
  static const int lo_cols[4] = {0, 1, 4, 5}, hi_cols[4] = {2, 3, 6, 7};
  for (size_t y = 0; y < H / 2; y += 8)
  {
    if (y * 2 < startY)
      continue;
    if (y * 2 > endY)
      break;
    for (size_t pr = 0; pr < W / 16; pr++)
    {
      uint8_t *base = to + y * W + pr * 128;
      for (int ab = 0; ab < 2; ab++)
      {
        float blk[64];
        sse_load(from + y * W + pr * 16 + ab * 8, W, blk);
        rows(blk, K_SSE); transpose(blk); rows(blk, K_SSE);
        for (int i8 = 0; i8 < 8; i8++)
          for (int k = 0; k < 4; k++)
          {
            base[i8 * 8 + ab * 4 + k] = sse_quant(blk[i8 * 8 + lo_cols[k]], q[i8 * 8 + lo_cols[k]]);
            /* the reference has no bounds check here; the oracle refuses to write
             * outside the W*H buffer (only reachable when H == 8). */
            if ((size_t)(base + 128 + i8 * 8 + ab * 4 + k - to) < W * H)
              base[128 + i8 * 8 + ab * 4 + k] = sse_quant(blk[i8 * 8 + hi_cols[k]], q[i8 * 8 + hi_cols[k]]);
          }
      }
    }
  }
  return 0;
}

/* ------------------------------------------------------------ B4 / B5 ----- */
/* scalar tiers: px / 255.f (:222, :343), K_TRUE, qs[i] = 1.f/(lut[i]*0.95f) (:192-209),
 * byte = (uint8_t)roundf(clamp(f*qs[i] + 127.0f/255.0f, 0, 1) * 255.f) (:245, :362),
 * _clamp(v,min,max) = v > min ? (v < max ? v : max) : min (:50-54). */
static uint8_t scalar_quant(float f, float qs)
{
  const float subtract = 127.0f / 255.0f;
  float v = (f * qs) + subtract;
  v = v > 0.f ? (v < 1.f ? v : 1.f) : 0.f;
  return (uint8_t)roundf(v * 255.f);
}

static void scalar_load(const uint8_t *src, size_t pitch, float *blk)
{
  for (int r = 0; r < 8; r++)
    for (int c = 0; c < 8; c++)
      blk[r * 8 + c] = (float)src[r * pitch + c] / 255.f;
}

static void make_qs(const float *lut, float *qs)
{
  for (int i = 0; i < 64; i++)
    qs[i] = 1.f / (lut[i] * 0.95f);
}

/* simd_dct.cpp:177-298: T, K_TRUE rows, T, K_TRUE rows; layout as B2. */
int orc_stereo_scalar(const uint8_t *from, uint8_t *to, const float *lut, size_t W, size_t H, size_t startY, size_t endY)
{
  const int e = check_args(from, to, W, H, 8);
  if (e)
    return e;
  if (H % 16 != 0)
    return 2; /* second image starts at row H/2 (:292): the reference over-reads otherwise */
  float qs[64];
  make_qs(lut, qs);
  const size_t plane = (W * H) / 64;
  size_t pos = 0;
  for (size_t y = 0; y < H / 2; y += 8)
  {
    if (y * 2 < startY)
    {
      pos += W / 4;
      continue;
    }
    if (y * 2 > endY)
      break;
    for (int eye = 0; eye < 2; eye++)
      for (size_t bx = 0; bx < W / 8; bx++)
      {
        float blk[64];
        scalar_load(from + (size_t)eye * (W * H / 2) + y * W + bx * 8, W, blk);
        transpose(blk); rows(blk, K_TRUE); transpose(blk); rows(blk, K_TRUE);
        for (int i = 0; i < 64; i++)
          to[plane * i + pos] = scalar_quant(blk[i], qs[i]);
        pos++;
      }
  }
  return 0;
}

/* simd_dct.cpp:300-395: K_TRUE rows, T, K_TRUE rows (stored transposed, u*8+v);
 * 64 bytes per block at y*W + bx*64; range test WITHOUT the x2 (:377-384). */
int orc_encq_scalar(const uint8_t *from, uint8_t *to, const float *lut, size_t W, size_t H, size_t startY, size_t endY)
{
  const int e = check_args(from, to, W, H, 8);
  if (e)
    return e;
  float qs[64];
  make_qs(lut, qs);
  for (size_t y = 0; y < H / 2; y += 8)
  {
    if (y < startY)
      continue;
    if (y > endY)
      break;
    for (size_t bx = 0; bx < W / 8; bx++)
    {
      float blk[64];
      scalar_load(from + y * W + bx * 8, W, blk);
      rows(blk, K_TRUE); transpose(blk); rows(blk, K_TRUE);
      for (int i = 0; i < 64; i++)
        to[y * W + bx * 64 + i] = scalar_quant(blk[i], qs[i]);
    }
  }
  return 0;
}

/* ------------------------------------------------- engine-own [unpinned] -- */
/* Definition of the engine's own transforms (what simd_dct_amd/csrc implements):
 *   raw forward  R = AANcols(AANrows(x))                      (no scaling inside)
 *   fwd  i16 : c[i] = quant_i16(R[i], QF[i]) = sat_i16(rne(R[i] * QF[i])) with one rounding,  QF[i] = fwdtab[i]  (no table)
 *                                                   QF[i] = (1.0f/lut[i]) * fwdtab[i]  (table)
 *   inv  i16 : z[i] = (float)c[i] * DQ[i],          DQ[i] = invtab[i] | lut[i] * invtab[i]
 *              x = sat_i16(rne(AANrows^-1(AANcols^-1(z))))
 *   roundtrip, no table : x' = sat_i16(rne(AANinv(R) * (1/64)))   (fwdtab*invtab == 1/64; a power
 *              of two commutes with every rounding, so it is applied once at the end)
 *   roundtrip, table    : z[i] = (float)quant_i16(R[i], QF[i]) * DQ[i], then as inv
 *   f32      : fwd out = R[i]*fwdtab[i];  inv z = in[i]*invtab[i]
 */
/* The engine-own QUANTISER (round 6): c = sat_i16(rne(y * qf)) with ONE rounding.  fmaf(y, qf, 1.5 * 2^23) is the exact product rounded
 * to nearest-even in units of 1 (for |y qf| < 2^22; beyond that the clamp decides); the clamp works on the biased value, whose bounds
 * 1.5 * 2^23 - 32768 and + 32767 are exact floats.  The engine's kernels perform the same three operations (mdct_kernels.hip:
 * quant_i16_bits).  Rounds 1-5 rounded the product to a float first and that float to an integer (two roundings). */
static inline __attribute__((always_inline)) int16_t quant_i16(float y, float qf)
{
  const float magic = 12582912.0f, lo = 12582912.0f - 32768.0f, hi = 12582912.0f + 32767.0f;
  float t = fmaf(y, qf, magic);
  if (t != t)
    return 0;
  t = t < lo ? lo : (t > hi ? hi : t);
  return (int16_t)(t - magic);
}

/* The 8-bit output stage (round 6): px = sat_u8(rne(x)) -- round to nearest even, saturate to [0, 255], NaN -> 0, which is what the
 * engine's one-instruction-per-pixel convert does.  The level shift of the OUTPUT (+ 128) is not added to the rounded pixel any more: it
 * rides in the DC term, z00 + shift before the inverse transform (a constant plane is exactly the DC term of the AAN inverse), so the
 * shifted value takes part in the butterflies' roundings. */
static inline __attribute__((always_inline)) uint8_t sat_u8_rne(float x)
{
  const float r = rintf(x);
  if (r != r)
    return 0;
  return (uint8_t)(r < 0.f ? 0.f : (r > 255.f ? 255.f : r));
}

/* the two definitions above, exported for the tests that hold them against exact arithmetic (tests/test_oracle.py) */
void orc_quant_i16(const float *y, const float *qf, int16_t *out, size_t n)
{
  for (size_t i = 0; i < n; i++)
    out[i] = quant_i16(y[i], qf[i]);
}
void orc_sat_u8_rne(const float *x, uint8_t *out, size_t n)
{
  for (size_t i = 0; i < n; i++)
    out[i] = sat_u8_rne(x[i]);
}

static int16_t sat_i16_rne(float v)
{
  const float r = rintf(v);
  if (r != r)
    return 0;
  if (r <= -32768.f)
    return INT16_MIN;
  if (r >= 32767.f)
    return INT16_MAX;
  return (int16_t)r;
}

static int own_args(const void *from, const void *to, size_t pi, size_t po, size_t W, size_t H, size_t by0, size_t by1)
{
  if (from == NULL || to == NULL)
    return 1;
  if (W % 8 != 0 || H % 8 != 0)
    return 2;
  if (pi < W || po < W || by0 > by1 || by1 > H / 8)
    return 1;
  return 0;
}

static void own_tables(const float *lut, float *qf, float *dq)
{
  float ft[64], it[64];
  orc_aan_tables(ft, it);
  for (int i = 0; i < 64; i++)
  {
    qf[i] = lut ? (1.0f / lut[i]) * ft[i] : ft[i];
    dq[i] = lut ? lut[i] * it[i] : it[i];
  }
}

static inline __attribute__((always_inline)) void raw_fwd(float *blk)
{
  for (int r = 0; r < 8; r++)
    aan_fwd8(blk + r * 8, 1);
  for (int c = 0; c < 8; c++)
    aan_fwd8(blk + c, 8);
}

static inline __attribute__((always_inline)) void raw_inv(float *blk)
{
  for (int c = 0; c < 8; c++)
    aan_inv8(blk + c, 8);
  for (int r = 0; r < 8; r++)
    aan_inv8(blk + r * 8, 1);
}

#define FOR_BLOCKS for (size_t by = by0; by < by1; by++) for (size_t bx = 0; bx < W / 8; bx++)
#define LOAD_I16(blk) for (int r = 0; r < 8; r++) for (int c = 0; c < 8; c++) blk[r * 8 + c] = (float)from[(by * 8 + r) * pi + bx * 8 + c]
#define AT(r, c) to[(by * 8 + (r)) * po + bx * 8 + (c)]

ORC_FMA_CLONES int orc_fwd_i16(const int16_t *from, int16_t *to, size_t pi, size_t po, const float *lut, size_t W, size_t H, size_t by0, size_t by1)
{
  const int e = own_args(from, to, pi, po, W, H, by0, by1);
  if (e)
    return e;
  float qf[64], dq[64];
  own_tables(lut, qf, dq);
  FOR_BLOCKS
  {
    float blk[64];
    LOAD_I16(blk);
    raw_fwd(blk);
    for (int i = 0; i < 64; i++)
      AT(i >> 3, i & 7) = quant_i16(blk[i], qf[i]);
  }
  return 0;
}

ORC_FMA_CLONES int orc_inv_i16(const int16_t *from, int16_t *to, size_t pi, size_t po, const float *lut, size_t W, size_t H, size_t by0, size_t by1)
{
  const int e = own_args(from, to, pi, po, W, H, by0, by1);
  if (e)
    return e;
  float qf[64], dq[64];
  own_tables(lut, qf, dq);
  FOR_BLOCKS
  {
    float blk[64];
    LOAD_I16(blk);
    for (int i = 0; i < 64; i++)
      blk[i] = blk[i] * dq[i];
    raw_inv(blk);
    for (int i = 0; i < 64; i++)
      AT(i >> 3, i & 7) = sat_i16_rne(blk[i]);
  }
  return 0;
}

ORC_FMA_CLONES int orc_roundtrip_i16(const int16_t *from, int16_t *to, size_t pi, size_t po, const float *lut, size_t W, size_t H, size_t by0, size_t by1)
{
  const int e = own_args(from, to, pi, po, W, H, by0, by1);
  if (e)
    return e;
  float qf[64], dq[64];
  own_tables(lut, qf, dq);
  FOR_BLOCKS
  {
    float blk[64];
    LOAD_I16(blk);
    raw_fwd(blk);
    if (lut)
      for (int i = 0; i < 64; i++)
        blk[i] = (float)quant_i16(blk[i], qf[i]) * dq[i];
    raw_inv(blk);
    for (int i = 0; i < 64; i++)
      AT(i >> 3, i & 7) = sat_i16_rne(lut ? blk[i] : blk[i] * 0.015625f);
  }
  return 0;
}

/* 8-bit pixels <-> int16 coefficients (the JPEG-style pair).  level_shift != 0 centres the
 * pixels on zero (x - 128) on the way in and adds 128 back on the way out; pixels saturate to
 * [0, 255].  Same AAN arithmetic and tables as the int16 entry points. */
ORC_FMA_CLONES int orc_fwd_u8_i16(const uint8_t *from, int16_t *to, size_t pi, size_t po, const float *lut, int level_shift, size_t W, size_t H, size_t by0, size_t by1)
{
  const int e = own_args(from, to, pi, po, W, H, by0, by1);
  if (e)
    return e;
  float qf[64], dq[64];
  own_tables(lut, qf, dq);
  FOR_BLOCKS
  {
    float blk[64];
    for (int r = 0; r < 8; r++)
      for (int c = 0; c < 8; c++)
        blk[r * 8 + c] = (float)from[(by * 8 + r) * pi + bx * 8 + c] - (level_shift ? 128.0f : 0.0f);
    raw_fwd(blk);
    for (int i = 0; i < 64; i++)
      AT(i >> 3, i & 7) = quant_i16(blk[i], qf[i]);
  }
  return 0;
}

ORC_FMA_CLONES int orc_inv_i16_u8(const int16_t *from, uint8_t *to, size_t pi, size_t po, const float *lut, int level_shift, size_t W, size_t H, size_t by0, size_t by1)
{
  const int e = own_args(from, to, pi, po, W, H, by0, by1);
  if (e)
    return e;
  float qf[64], dq[64];
  own_tables(lut, qf, dq);
  FOR_BLOCKS
  {
    float blk[64];
    LOAD_I16(blk);
    for (int i = 0; i < 64; i++)
      blk[i] = blk[i] * dq[i];
    blk[0] = blk[0] + (level_shift ? 128.0f : 0.0f); /* the output's level shift rides in the DC term (sat_u8_rne above) */
    raw_inv(blk);
    for (int i = 0; i < 64; i++)
      AT(i >> 3, i & 7) = sat_u8_rne(blk[i]);
  }
  return 0;
}

/* Fused 8-bit round trip (BASELINE.json configs[2] as SURVEY.md 8(d) states it: u8 planes in, u8 planes out): per block exactly
 * orc_fwd_u8_i16 followed by orc_inv_i16_u8 with the same table and level shift -- the int16 coefficients exist only in between
 * (tests/test_oracle.py checks this function against that composition).  pitches in bytes. */
ORC_FMA_CLONES int orc_roundtrip_u8(const uint8_t *from, uint8_t *to, size_t pi, size_t po, const float *lut, int level_shift, size_t W, size_t H, size_t by0, size_t by1)
{
  const int e = own_args(from, to, pi, po, W, H, by0, by1);
  if (e)
    return e;
  float qf[64], dq[64];
  own_tables(lut, qf, dq);
  FOR_BLOCKS
  {
    float blk[64];
    for (int r = 0; r < 8; r++)
      for (int c = 0; c < 8; c++)
        blk[r * 8 + c] = (float)from[(by * 8 + r) * pi + bx * 8 + c] - (level_shift ? 128.0f : 0.0f);
    raw_fwd(blk);
    for (int i = 0; i < 64; i++)
      blk[i] = (float)quant_i16(blk[i], qf[i]) * dq[i];
    blk[0] = blk[0] + (level_shift ? 128.0f : 0.0f);
    raw_inv(blk);
    for (int i = 0; i < 64; i++)
      AT(i >> 3, i & 7) = sat_u8_rne(blk[i]);
  }
  return 0;
}

ORC_FMA_CLONES int orc_fwd_f32(const float *from, float *to, size_t pi, size_t po, size_t W, size_t H, size_t by0, size_t by1)
{
  const int e = own_args(from, to, pi, po, W, H, by0, by1);
  if (e)
    return e;
  float ft[64], it[64];
  orc_aan_tables(ft, it);
  FOR_BLOCKS
  {
    float blk[64];
    for (int r = 0; r < 8; r++)
      memcpy(blk + r * 8, from + (by * 8 + r) * pi + bx * 8, 8 * sizeof(float));
    raw_fwd(blk);
    for (int i = 0; i < 64; i++)
      AT(i >> 3, i & 7) = blk[i] * ft[i];
  }
  return 0;
}

ORC_FMA_CLONES int orc_inv_f32(const float *from, float *to, size_t pi, size_t po, size_t W, size_t H, size_t by0, size_t by1)
{
  const int e = own_args(from, to, pi, po, W, H, by0, by1);
  if (e)
    return e;
  float ft[64], it[64];
  orc_aan_tables(ft, it);
  FOR_BLOCKS
  {
    float blk[64];
    for (int r = 0; r < 8; r++)
      memcpy(blk + r * 8, from + (by * 8 + r) * pi + bx * 8, 8 * sizeof(float));
    for (int i = 0; i < 64; i++)
      blk[i] = blk[i] * it[i];
    raw_inv(blk);
    for (int i = 0; i < 64; i++)
      AT(i >> 3, i & 7) = blk[i];
  }
  return 0;
}

int orc_fwd_f64ref(const float *from, double *to, size_t pi, size_t po, size_t W, size_t H, size_t by0, size_t by1)
{
  const int e = own_args(from, to, pi, po, W, H, by0, by1);
  if (e)
    return e;
  double c[8][8]; /* c[k][n] = s_k cos((2n+1) k pi / 16), orthonormal DCT-II */
  const double pi_ = 3.14159265358979323846264338327950288;
  for (int k = 0; k < 8; k++)
    for (int n = 0; n < 8; n++)
      c[k][n] = (k == 0 ? sqrt(1.0 / 8.0) : 0.5) * cos((2 * n + 1) * k * pi_ / 16.0);
  for (size_t by = by0; by < by1; by++)
    for (size_t bx = 0; bx < W / 8; bx++)
    {
      double t[8][8];
      for (int r = 0; r < 8; r++)
        for (int u = 0; u < 8; u++)
        {
          double s = 0;
          for (int n = 0; n < 8; n++)
            s += c[u][n] * (double)from[(by * 8 + r) * pi + bx * 8 + n];
          t[r][u] = s;
        }
      for (int v = 0; v < 8; v++)
        for (int u = 0; u < 8; u++)
        {
          double s = 0;
          for (int n = 0; n < 8; n++)
            s += c[v][n] * t[n][u];
          to[(by * 8 + v) * po + bx * 8 + u] = s;
        }
    }
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * Stages either side of the codec core [no reference counterpart: the
 * reference's pipeline ends at the reorder store, simd_dct.cpp:2221-2230, :1034-1052].
 * Published definitions restated here: the zig-zag scan of ITU-T T.81 (JPEG) Figure A.6 /
 * Figure 5, generated by walking the anti-diagonals (the product carries the table as
 * literals, so the two check each other), run/level pairs as in T.81 F.1.2.2 (a run is the
 * number of zero coefficients preceding a non-zero one in zig-zag order; trailing zeros are the
 * end of block), and 4:2:0 chroma siting as in JFIF (2x2 box average, centred).
 * ---------------------------------------------------------------------------------------- */
void orc_zigzag_table(uint8_t *zz)
{ /* zz[k] = natural index v*8+u of the k-th coefficient of the scan */
  int v = 0, u = 0;
  for (int k = 0; k < 64; k++)
  {
    zz[k] = (uint8_t)(v * 8 + u);
    if ((v + u) % 2 == 0)
    { /* moving up-right */
      if (u == 7)
        v++;
      else if (v == 0)
        u++;
      else
      {
        v--;
        u++;
      }
    }
    else
    { /* moving down-left */
      if (v == 7)
        u++;
      else if (u == 0)
        v++;
      else
      {
        v++;
        u--;
      }
    }
  }
}

/* one block's 64 values in natural order -> scan order, then (run, level) pairs.
 * runs == NULL: plain zig-zag scan (levels = all 64 values in scan order). */
static void scan_block(const int *nat, int16_t *levels, uint8_t *runs, uint8_t *count)
{
  uint8_t zz[64];
  orc_zigzag_table(zz);
  if (!runs)
  {
    for (int k = 0; k < 64; k++)
      levels[k] = (int16_t)nat[zz[k]];
    return;
  }
  int n = 0, run = 0;
  memset(levels, 0, 64 * sizeof(int16_t));
  memset(runs, 0, 64);
  for (int k = 0; k < 64; k++)
  {
    const int c = nat[zz[k]];
    if (c != 0)
    {
      levels[n] = (int16_t)c;
      runs[n] = (uint8_t)run;
      n++;
      run = 0;
    }
    else
      run++;
  }
  *count = (uint8_t)n;
}

/* int16 coefficient plane (coefficient (v,u) of block (by,bx) at (by*8+v, bx*8+u)) -> records of
 * block by*bpr+bx: levels[blk*64 ..], runs[blk*64 ..], counts[blk] */
int orc_zigzag_rle_i16(const int16_t *coef, size_t pitch, size_t W, size_t H, size_t by0, size_t by1, int16_t *levels, uint8_t *runs, uint8_t *counts)
{
  if (!coef || !levels || (runs && !counts))
    return 1;
  if (W % 8 || H % 8)
    return 2;
  if (pitch < W || by0 > by1 || by1 > H / 8)
    return 1;
  const size_t bpr = W / 8;
  for (size_t by = by0; by < by1; by++)
    for (size_t bx = 0; bx < bpr; bx++)
    {
      int nat[64];
      for (int i = 0; i < 64; i++)
        nat[i] = coef[(by * 8 + (size_t)(i >> 3)) * pitch + bx * 8 + (size_t)(i & 7)];
      const size_t blk = by * bpr + bx;
      scan_block(nat, levels + blk * 64, runs ? runs + blk * 64 : NULL, counts ? counts + blk : NULL);
    }
  return 0;
}

/* the reference's q32 layout (simd_dct.cpp:2221-2230): byte [coef*8 + b] of the 512-byte group; the
 * stored byte carries the +127 bias of :2224, so level = byte - 127 */
int orc_zigzag_rle_q32(const uint8_t *q32, size_t W, size_t H, size_t by0, size_t by1, int16_t *levels, uint8_t *runs, uint8_t *counts)
{
  if (!q32 || !levels || (runs && !counts))
    return 1;
  if (W % 64 || H % 8)
    return 2;
  if (by0 > by1 || by1 > H / 8)
    return 1;
  const size_t bpr = W / 8;
  for (size_t by = by0; by < by1; by++)
    for (size_t bx = 0; bx < bpr; bx++)
    {
      int nat[64];
      const uint8_t *grp = q32 + by * 8 * W + (bx / 8) * 512;
      for (int i = 0; i < 64; i++)
        nat[i] = (int)grp[(size_t)i * 8 + bx % 8] - 127;
      const size_t blk = by * bpr + bx;
      scan_block(nat, levels + blk * 64, runs ? runs + blk * 64 : NULL, counts ? counts + blk : NULL);
    }
  return 0;
}

/* interleaved 8-bit Y Cb Cr (3 bytes per pixel) -> three int16 planes, level-shifted by -128:
 * Y at full resolution, Cb / Cr subsampled 2x2 by the rounded box average (a+b+c+d+2) >> 2. */
int orc_split420_u8(const uint8_t *ycc, size_t pitch, size_t W, size_t H, int16_t *y, int16_t *cb, int16_t *cr, size_t pitch_y, size_t pitch_c)
{
  if (!ycc || !y || !cb || !cr)
    return 1;
  if (W % 16 || H % 16)
    return 2;
  if (pitch < 3 * W || pitch_y < W || pitch_c < W / 2)
    return 1;
  for (size_t r = 0; r < H; r++)
    for (size_t c = 0; c < W; c++)
      y[r * pitch_y + c] = (int16_t)((int)ycc[r * pitch + 3 * c] - 128);
  for (size_t r = 0; r < H / 2; r++)
    for (size_t c = 0; c < W / 2; c++)
      for (int k = 1; k <= 2; k++)
      {
        const uint8_t *p0 = ycc + (2 * r) * pitch + 3 * (2 * c) + (size_t)k, *p1 = p0 + pitch;
        const int s = (int)p0[0] + (int)p0[3] + (int)p1[0] + (int)p1[3];
        (k == 1 ? cb : cr)[r * pitch_c + c] = (int16_t)(((s + 2) >> 2) - 128);
      }
  return 0;
}

/* the same from the stereo layout (64 coefficient planes, simd_dct.cpp:1061-1099; block index = stream
 * position (by*2+eye)*bpr + bx, ranges in units of 16 pixel rows) and from the scalar encq tier's block
 * layout (:347-362, coefficient (v,u) at u*8+v); both carry the +127 bias */
int orc_zigzag_rle_u8(const uint8_t *coef, int layout, size_t W, size_t H, size_t by0, size_t by1, int16_t *levels, uint8_t *runs, uint8_t *counts)
{
  if (layout == 0)
    return orc_zigzag_rle_q32(coef, W, H, by0, by1, levels, runs, counts);
  if (!coef || !levels || (runs && !counts))
    return 1;
  if (layout != 1 && layout != 2)
    return 2;
  const size_t ymul = layout == 1 ? 16 : 8;
  if (W % (layout == 1 ? 16 : 8) || H % ymul)
    return 2;
  if (by0 > by1 || by1 > H / ymul)
    return 1;
  const size_t bpr = (layout == 1 ? 2 : 1) * (W / 8), plane = W * H / 64;
  for (size_t blk = by0 * bpr; blk < by1 * bpr; blk++)
  {
    int nat[64];
    for (int i = 0; i < 64; i++)
      nat[i] = layout == 1 ? (int)coef[(size_t)i * plane + blk] - 127 : (int)coef[blk * 64 + (size_t)((i & 7) * 8 + (i >> 3))] - 127;
    scan_block(nat, levels + blk * 64, runs ? runs + blk * 64 : NULL, counts ? counts + blk : NULL);
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * Baseline Huffman coding of the run/level records (SURVEY.md 8 f4 "entropy stage") [no reference
 * counterpart].  Restates ITU-T T.81: code construction Annex C, coding procedures F.1.2.1 (DC:
 * DIFF category + amplitude) and F.1.2.2 (AC: RRRRSSSS with ZRL and EOB), the typical tables of
 * Annex K.3.3 (Tables K.3-K.6).  One byte-aligned, independently decodable segment per block row:
 * the DC predictor starts at 0 in every row (a restart interval of sizeX/8 blocks, E.1.4) and the
 * last byte is padded with 1-bits (F.1.2.3).  Bytes are NOT stuffed (B.1.1.5): whoever writes the
 * file inserts 0x00 after every 0xFF and the RSTm markers between rows.
 * ---------------------------------------------------------------------------------------- */
static const uint8_t kDcLumaBits[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
static const uint8_t kDcChromaBits[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
static const uint8_t kDcVals[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
static const uint8_t kAcLumaBits[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
static const uint8_t kAcLumaVals[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1,
    0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37,
    0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a,
    0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3,
    0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3,
    0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
static const uint8_t kAcChromaBits[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
static const uint8_t kAcChromaVals[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1,
    0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36,
    0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69,
    0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a,
    0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca,
    0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};

/* BITS (16) and HUFFVAL of the table, as a DHT segment carries them.  which: 0 DC luma, 1 AC luma, 2 DC chroma, 3 AC chroma */
int orc_huffman_spec(int which, uint8_t *bits16, uint8_t *vals, int *nvals)
{
  const uint8_t *b = which == 0 ? kDcLumaBits : (which == 1 ? kAcLumaBits : (which == 2 ? kDcChromaBits : kAcChromaBits));
  const uint8_t *v = which == 1 ? kAcLumaVals : (which == 3 ? kAcChromaVals : kDcVals);
  if (which < 0 || which > 3)
    return 1;
  int n = 0;
  for (int i = 0; i < 16; i++)
  {
    bits16[i] = b[i];
    n += b[i];
  }
  memcpy(vals, v, (size_t)n);
  *nvals = n;
  return 0;
}

/* Annex C: code[sym] and size[sym] from BITS / HUFFVAL; table entry = size << 16 | code (0 = symbol absent) */
static void huff_build(const uint8_t *bits, const uint8_t *vals, uint32_t *tab256)
{
  memset(tab256, 0, 256 * sizeof(uint32_t));
  uint32_t code = 0;
  int k = 0;
  for (int len = 1; len <= 16; len++)
  {
    for (int i = 0; i < bits[len - 1]; i++)
      tab256[vals[k++]] = ((uint32_t)len << 16) | code++;
    code <<= 1;
  }
}

void orc_huffman_tables(int chroma, uint32_t *dc256, uint32_t *ac256)
{
  huff_build(chroma ? kDcChromaBits : kDcLumaBits, kDcVals, dc256);
  huff_build(chroma ? kAcChromaBits : kAcLumaBits, chroma ? kAcChromaVals : kAcLumaVals, ac256);
}

typedef struct
{
  uint8_t *p;
  uint64_t acc; /* bits not yet written, right-aligned */
  int n;
  size_t bytes;
} bitw;

static void put_bits(bitw *w, uint32_t code, int len)
{
  w->acc = (w->acc << len) | (code & ((1u << len) - 1u));
  w->n += len;
  while (w->n >= 8)
  {
    w->p[w->bytes++] = (uint8_t)(w->acc >> (w->n - 8));
    w->n -= 8;
  }
}

static int bit_size(int v)
{
  int a = v < 0 ? -v : v, s = 0;
  while (a)
  {
    s++;
    a >>= 1;
  }
  return s;
}

int orc_huffman_rows(const int16_t *levels, const uint8_t *runs, const uint8_t *counts, size_t W, size_t H, size_t by0, size_t by1, int chroma, uint8_t *out,
                     size_t seg_stride, uint32_t *seg_bytes)
{
  if (!levels || !runs || !counts || !out || !seg_bytes)
    return 1;
  if (W % 8 || H % 8)
    return 2;
  const size_t bpr = W / 8;
  if (by0 > by1 || by1 > H / 8 || seg_stride < bpr * 208 + 8)
    return 1;
  uint32_t dct[256], act[256];
  orc_huffman_tables(chroma, dct, act);
  for (size_t by = by0; by < by1; by++)
  {
    bitw w = {out + by * seg_stride, 0, 0, 0};
    int pred = 0;
    for (size_t bx = 0; bx < bpr; bx++)
    {
      const size_t blk = by * bpr + bx;
      const int16_t *lv = levels + blk * 64;
      const uint8_t *rn = runs + blk * 64;
      int n = counts[blk] > 64 ? 64 : counts[blk];
      int i = 0, dc = 0;
      if (n > 0 && rn[0] == 0)
        dc = lv[i++]; /* the first pair sits at scan position 0 */
      { /* not a block (a zero AC level, or positions past 63): coded as its DC coefficient alone, so that the
           worst case of F.1.2 (which sizes seg_stride) holds for any input */
        int pp = i ? 0 : -1, bad = 0;
        for (int k = i; k < n; k++)
        {
          pp += rn[k] + 1;
          bad |= lv[k] == 0;
        }
        if (bad || pp > 63)
          n = i;
      }
      int diff = dc - pred;
      pred = dc;
      diff = diff > 2047 ? 2047 : (diff < -2047 ? -2047 : diff); /* 8-bit baseline: categories 0..11 (F.1.2.1.1) */
      int s = bit_size(diff);
      put_bits(&w, dct[s] & 0xFFFF, (int)(dct[s] >> 16));
      if (s)
        put_bits(&w, (uint32_t)(diff < 0 ? diff - 1 : diff), s);
      int q = 0, p = (n > 0 && rn[0] == 0) ? 0 : -1; /* q: previous coded position, p: position of pair i-1 */
      for (; i < n; i++)
      {
        p += rn[i] + 1;
        int r = p - q - 1, l = lv[i];
        q = p;
        l = l > 1023 ? 1023 : (l < -1023 ? -1023 : l); /* categories 1..10 (F.1.2.2.1) */
        for (; r > 15; r -= 16)
          put_bits(&w, act[0xF0] & 0xFFFF, (int)(act[0xF0] >> 16)); /* ZRL */
        s = bit_size(l);
        const uint32_t e = act[(r << 4) | s];
        put_bits(&w, e & 0xFFFF, (int)(e >> 16));
        put_bits(&w, (uint32_t)(l < 0 ? l - 1 : l), s);
      }
      if (q < 63)
        put_bits(&w, act[0x00] & 0xFFFF, (int)(act[0x00] >> 16)); /* EOB */
    }
    if (w.n)
      put_bits(&w, 0xFF, 8 - w.n); /* pad with 1-bits */
    seg_bytes[by] = (uint32_t)w.bytes;
  }
  return 0;
}

/* the row segments -> one stuffed scan with RSTm between the rows (ITU-T T.81 B.1.1.5, E.1.4); checker of
 * mdct_jpeg_pack_rows.  A row that would end beyond `capacity` is not written. */
int orc_jpeg_pack_rows(const uint8_t *seg, const uint32_t *seg_bytes, size_t seg_stride, size_t n_rows, int first_rst, uint8_t *out, size_t capacity, uint64_t *row_off)
{
  if (!seg || !seg_bytes || !out || !row_off)
    return 1;
  uint64_t pos = 0;
  for (size_t r = 0; r < n_rows; r++)
  {
    const uint8_t *p = seg + r * seg_stride;
    const uint32_t nb = seg_bytes[r] > seg_stride ? (uint32_t)seg_stride : seg_bytes[r]; /* a length beyond the stride is not a row */
    uint64_t len = nb;
    for (uint32_t i = 0; i < nb; i++)
      len += p[i] == 0xFF;
    if (r + 1 < n_rows)
      len += 2;
    row_off[r] = pos;
    if (pos + len <= capacity)
    {
      uint8_t *o = out + pos;
      for (uint32_t i = 0; i < nb; i++)
      {
        *o++ = p[i];
        if (p[i] == 0xFF)
          *o++ = 0;
      }
      if (r + 1 < n_rows)
      {
        *o++ = 0xFF;
        *o++ = (uint8_t)(0xD0 + ((first_rst + r) & 7));
      }
    }
    pos += len;
  }
  row_off[n_rows] = pos;
  return 0;
}
