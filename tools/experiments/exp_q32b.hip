// exp_q32b.hip -- round-2 A/B harness for the q32 kernel (u8 -> u8, 8-block interleave, AVX profile).
// Every variant must reproduce the product kernel's bytes exactly (checked before timing).
//   v_not      drop the 64 per-coefficient integer +127: quantise -v with the magic constant 1.5*2^23+128
//              (low byte = rne(-v)+128) and complement the packed dwords after the LDS reorder (4 v_not)
//   v_wide     v_not + 16 B/lane loads (two rows per 1 KiB wave load) and v_permlane32_swap
//   v_pipe     v_wide + persistent waves with register prefetch of the next tile
//   v_dma      v_not + persistent waves, next tile prefetched by LDS-DMA (global_load_lds_dwordx4)
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -std=c++17 -Iinclude -Isimd_dct_amd/csrc tools/exp_q32b.hip -o tools/exp_q32b
#include "../simd_dct_amd/csrc/mdct_kernels.hip"
#include "scalar_forms.h"
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <functional>
#include <vector>
using namespace mdct;

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));

// rows then columns (K_AVX), quantise -v: out word's low byte = ~(reference byte)
__device__ __forceinline__ void transform_quant_not(const DctConsts &C, float (&b)[8][8], const QuantTable &nq, float magicC, uint32_t (&out)[64])
{
  pass_rows<K_AVX>(C, b);
  pass_cols<K_AVX>(C, b);
#pragma unroll
  for (int s = 0; s < 64; s++)
    out[s] = __float_as_uint(__builtin_amdgcn_fmed3f(b[s >> 3][s & 7] * nq.q[s], -128.0f, 127.0f) + magicC);
}

__device__ __forceinline__ void rows_to_float(const uint2 (&rows)[8], float (&b)[8][8])
{
#pragma unroll
  for (int r = 0; r < 8; r++)
  {
    b[r][0] = ubyte_to_float<0>(rows[r].x); b[r][1] = ubyte_to_float<1>(rows[r].x);
    b[r][2] = ubyte_to_float<2>(rows[r].x); b[r][3] = ubyte_to_float<3>(rows[r].x);
    b[r][4] = ubyte_to_float<0>(rows[r].y); b[r][5] = ubyte_to_float<1>(rows[r].y);
    b[r][6] = ubyte_to_float<2>(rows[r].y); b[r][7] = ubyte_to_float<3>(rows[r].y);
  }
}

// stage the wave's 64 x 64 bytes as rows [coef][block], read back 16 B per lane, complement, store
template <bool NOT>
__device__ __forceinline__ void reorder_store(uint8_t *wl, const uint32_t (&q)[64], uint32_t blk, uint32_t lane, uint8_t *outw)
{
#pragma unroll
  for (int c = 0; c < 64; c++)
    wl[c * kQ32RowStride + blk] = (uint8_t)q[c];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const uint32_t c2 = (lane & 31) * 2;
#pragma unroll
  for (int k = 0; k < 4; k++)
  {
    const uint32_t g = 2 * k + (lane >> 5);
    const uint2 lo = *reinterpret_cast<const uint2 *>(wl + c2 * kQ32RowStride + g * 8);
    const uint2 hi = *reinterpret_cast<const uint2 *>(wl + (c2 + 1) * kQ32RowStride + g * 8);
    u32x4_t v = {lo.x, lo.y, hi.x, hi.y};
    if (NOT)
      v = ~v;
    __builtin_nontemporal_store(v, reinterpret_cast<u32x4_t *>(outw + g * 512 + c2 * 8));
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); // the next tile's LDS writes stay behind these reads
  __builtin_amdgcn_wave_barrier();
}

// ---- v_not: product structure, cheaper quantiser
template <int MINW>
__global__ __launch_bounds__(256, MINW) void v_not(U8Args a, float magicC)
{
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t row = t / a.bpr, bx = t - row * a.bpr;
  const uint8_t *src = a.from + (size_t)(a.by0 + row) * 8 * a.pitch + (size_t)bx * 8;
  __shared__ __attribute__((aligned(16))) uint8_t lds[4][64 * kQ32RowStride];
  uint2 rows[8];
  load_block_rows(src, a.pitch, rows);
  float b[8][8];
  rows_to_float(rows, b);
  uint32_t q[64];
  transform_quant_not(a.consts, b, a.qt, magicC, q);
  reorder_store<true>(lds[threadIdx.x >> 6], q, lane, lane, a.to + ((size_t)a.by0 * a.bpr + (t - lane)) * 64);
}

// ---- wide loads: lanes 0..31 fetch 16 B of row 2k, lanes 32..63 of row 2k+1; one permlane32 swap per dword
// leaves row 2k in (x,y) and row 2k+1 in (z,w) for every lane; lane l<32 owns block 2l, lane 32+l block 2l+1
__device__ __forceinline__ void load_wide(const uint8_t *src16, size_t pitch, u32x4_t (&v)[4])
{
#pragma unroll
  for (int k = 0; k < 4; k++)
    v[k] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t *>(src16 + (size_t)(2 * k) * pitch));
}
__device__ __forceinline__ void wide_to_rows(const u32x4_t (&v)[4], uint2 (&rows)[8])
{
#pragma unroll
  for (int k = 0; k < 4; k++)
  {
    const auto s0 = __builtin_amdgcn_permlane32_swap(v[k].x, v[k].z, false, false);
    const auto s1 = __builtin_amdgcn_permlane32_swap(v[k].y, v[k].w, false, false);
    rows[2 * k] = make_uint2(s0[0], s1[0]);
    rows[2 * k + 1] = make_uint2(s0[1], s1[1]);
  }
}

template <int MINW>
__global__ __launch_bounds__(256, MINW) void v_wide(U8Args a, float magicC)
{
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  const uint32_t lane = threadIdx.x & 63, half = lane >> 5, l32 = lane & 31;
  const uint32_t t0 = t - lane; // the wave's first block; all 64 share a block row (bpr % 64 == 0)
  const uint32_t row = t0 / a.bpr, bx0 = t0 - row * a.bpr;
  const uint8_t *src16 = a.from + ((size_t)(a.by0 + row) * 8 + half) * a.pitch + (size_t)bx0 * 8 + l32 * 16;
  __shared__ __attribute__((aligned(16))) uint8_t lds[4][64 * kQ32RowStride];
  u32x4_t v[4];
  load_wide(src16, a.pitch, v);
  uint2 rows[8];
  wide_to_rows(v, rows);
  float b[8][8];
  rows_to_float(rows, b);
  uint32_t q[64];
  transform_quant_not(a.consts, b, a.qt, magicC, q);
  reorder_store<true>(lds[threadIdx.x >> 6], q, l32 * 2 + half, lane, a.to + ((size_t)a.by0 * a.bpr + t0) * 64);
}

// ---- persistent waves, register prefetch (wide loads): the raw bytes of tile i are dead after the swaps/converts
template <int MINW>
__global__ __launch_bounds__(256, MINW) void v_pipe(U8Args a, float magicC, uint32_t ntiles)
{
  __shared__ __attribute__((aligned(16))) uint8_t lds[4][64 * kQ32RowStride];
  const uint32_t lane = threadIdx.x & 63, half = lane >> 5, l32 = lane & 31;
  uint8_t *wl = lds[threadIdx.x >> 6];
  const uint32_t total_waves = gridDim.x * 4;
  uint32_t tile = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tile >= ntiles)
    return;
  auto src_of = [&](uint32_t tl) {
    const uint32_t t0 = tl * 64;
    const uint32_t row = t0 / a.bpr, bx0 = t0 - row * a.bpr;
    return a.from + ((size_t)(a.by0 + row) * 8 + half) * a.pitch + (size_t)bx0 * 8 + l32 * 16;
  };
  u32x4_t v[4];
  load_wide(src_of(tile), a.pitch, v);
  for (;;)
  {
    uint2 rows[8];
    wide_to_rows(v, rows);
    float b[8][8];
    rows_to_float(rows, b);
    const uint32_t next = tile + total_waves;
    const bool more = next < ntiles;
    if (more)
      load_wide(src_of(next), a.pitch, v);
    uint32_t q[64];
    transform_quant_not(a.consts, b, a.qt, magicC, q);
    reorder_store<true>(wl, q, l32 * 2 + half, lane, a.to + ((size_t)a.by0 * a.bpr + (size_t)tile * 64) * 64);
    if (!more)
      break;
    tile = next;
  }
}

// ---- persistent waves, LDS-DMA prefetch: tile i+1 lands in the wave's 4 KiB input buffer (rows of 512 B) while
// tile i is being transformed; no VGPR cost.  STAGES = 1: full 4.5 KiB output staging (<= 4 waves/SIMD by LDS);
// the input buffer is free again as soon as its 8 ds_read_b64 have returned.
template <int MINW>
__global__ __launch_bounds__(256, MINW) void v_dma(U8Args a, float magicC, uint32_t ntiles)
{
  __shared__ __attribute__((aligned(16))) uint8_t stage[4][64 * kQ32RowStride];
  __shared__ __attribute__((aligned(16))) uint8_t inbuf[4][8 * 512];
  const uint32_t lane = threadIdx.x & 63, half = lane >> 5, l32 = lane & 31;
  const uint32_t w = threadIdx.x >> 6;
  uint8_t *wl = stage[w];
  uint8_t *in = inbuf[w];
  const uint32_t total_waves = gridDim.x * 4;
  uint32_t tile = blockIdx.x * 4 + w;
  if (tile >= ntiles)
    return;
  auto issue = [&](uint32_t tl) {
    const uint32_t t0 = tl * 64;
    const uint32_t row = t0 / a.bpr, bx0 = t0 - row * a.bpr;
    const uint8_t *src16 = a.from + ((size_t)(a.by0 + row) * 8 + half) * a.pitch + (size_t)bx0 * 8 + l32 * 16;
#pragma unroll
    for (int k = 0; k < 4; k++) // rows 2k (lanes 0..31) and 2k+1 (lanes 32..63) land at in + k*1024 + lane*16
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src16 + (size_t)(2 * k) * a.pitch),
                                       (__attribute__((address_space(3))) void *)(in + k * 1024), 16, 0, 0);
  };
  issue(tile);
  for (;;)
  {
    // the 4 DMA pieces of this tile are older than the previous tile's 4 output stores: all but the 4 youngest done
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    uint2 rows[8];
#pragma unroll
    for (int r = 0; r < 8; r++)
      rows[r] = *reinterpret_cast<const uint2 *>(in + r * 512 + lane * 8);
    float b[8][8];
    rows_to_float(rows, b); // consumes the LDS reads
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const uint32_t next = tile + total_waves;
    const bool more = next < ntiles;
    if (more)
      issue(next);
    uint32_t q[64];
    transform_quant_not(a.consts, b, a.qt, magicC, q);
    reorder_store<true>(wl, q, lane, lane, a.to + ((size_t)a.by0 * a.bpr + (size_t)tile * 64) * 64);
    if (!more)
      break;
    tile = next;
  }
}


// ---- v_dma2: as v_dma, but (a) the wait for the prefetched tile sits right before the output stores (it has had the
// whole transform to land), so the loop top needs no VMEM wait and never waits for the previous tile's stores,
// (b) the output is staged in NPASS passes of 64/NPASS coefficients (LDS per wave 4096 + 4608/NPASS bytes), which
// lets 6..7 waves/SIMD fit the 160 KiB of LDS.
template <int NPASS>
__device__ __forceinline__ void reorder_store_passes(uint8_t *wl, const uint32_t (&q)[64], uint32_t blk, uint32_t lane, uint8_t *outw)
{
  constexpr int CP = 64 / NPASS;      // coefficients per pass
  constexpr int LPG = CP / 2;         // lanes per group (one coefficient pair each)
  constexpr int GPI = 64 / LPG;       // groups per store instruction
  constexpr int NS = CP / 16;         // 16-byte stores per lane per pass
  const uint32_t c2 = (lane % LPG) * 2;
#pragma unroll
  for (int p = 0; p < NPASS; p++)
  {
#pragma unroll
    for (int c = 0; c < CP; c++)
      wl[c * kQ32RowStride + blk] = (uint8_t)q[p * CP + c];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (p == 0)
      __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0): the next tile's LDS-DMA (issued before the transform) has landed
#pragma unroll
    for (int k = 0; k < NS; k++)
    {
      const uint32_t g = k * GPI + lane / LPG;
      const uint2 lo = *reinterpret_cast<const uint2 *>(wl + c2 * kQ32RowStride + g * 8);
      const uint2 hi = *reinterpret_cast<const uint2 *>(wl + (c2 + 1) * kQ32RowStride + g * 8);
      u32x4_t v = {lo.x, lo.y, hi.x, hi.y};
      v = ~v;
      __builtin_nontemporal_store(v, reinterpret_cast<u32x4_t *>(outw + g * 512 + (p * CP + c2) * 8));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

template <int MINW, int NPASS>
__global__ __launch_bounds__(256, MINW) void v_dma2(U8Args a, float magicC, uint32_t ntiles)
{
  __shared__ __attribute__((aligned(16))) uint8_t stage[4][(64 / NPASS) * kQ32RowStride];
  __shared__ __attribute__((aligned(16))) uint8_t inbuf[4][8 * 512];
  const uint32_t lane = threadIdx.x & 63, half = lane >> 5, l32 = lane & 31;
  const uint32_t w = threadIdx.x >> 6;
  uint8_t *wl = stage[w];
  uint8_t *in = inbuf[w];
  const uint32_t total_waves = gridDim.x * 4;
  uint32_t tile = blockIdx.x * 4 + w;
  if (tile >= ntiles)
    return;
  auto issue = [&](uint32_t tl) {
    const uint32_t t0 = tl * 64;
    const uint32_t row = t0 / a.bpr, bx0 = t0 - row * a.bpr;
    const uint8_t *src16 = a.from + ((size_t)(a.by0 + row) * 8 + half) * a.pitch + (size_t)bx0 * 8 + l32 * 16;
#pragma unroll
    for (int k = 0; k < 4; k++)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src16 + (size_t)(2 * k) * a.pitch),
                                       (__attribute__((address_space(3))) void *)(in + k * 1024), 16, 0, 0);
  };
  issue(tile);
  __builtin_amdgcn_s_waitcnt(0x0F70); // first tile: nothing to overlap with
  for (;;)
  {
    uint2 rows[8];
#pragma unroll
    for (int r = 0; r < 8; r++)
      rows[r] = *reinterpret_cast<const uint2 *>(in + r * 512 + lane * 8);
    float b[8][8];
    rows_to_float(rows, b);
    const uint32_t next = tile + total_waves;
    const bool more = next < ntiles;
    if (more)
      issue(next); // the converts above consumed every LDS read of the input buffer
    uint32_t q[64];
    transform_quant_not(a.consts, b, a.qt, magicC, q);
    reorder_store_passes<NPASS>(wl, q, lane, lane, a.to + ((size_t)a.by0 * a.bpr + (size_t)tile * 64) * 64);
    if (!more)
      break;
    tile = next;
  }
}

template <int MINW, int NPASS>
__global__ __launch_bounds__(256, MINW) void v_not_passes(U8Args a, float magicC)
{
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t row = t / a.bpr, bx = t - row * a.bpr;
  const uint8_t *src = a.from + (size_t)(a.by0 + row) * 8 * a.pitch + (size_t)bx * 8;
  __shared__ __attribute__((aligned(16))) uint8_t lds[4][(64 / NPASS) * kQ32RowStride];
  uint2 rows[8];
  load_block_rows(src, a.pitch, rows);
  float b[8][8];
  rows_to_float(rows, b);
  uint32_t q[64];
  transform_quant_not(a.consts, b, a.qt, magicC, q);
  reorder_store_passes<NPASS>(lds[threadIdx.x >> 6], q, lane, lane, a.to + ((size_t)a.by0 * a.bpr + (t - lane)) * 64);
}


// =====================================================================================================
// Packed-fp32 butterflies.  v_pk_add_f32 / v_pk_mul_f32 round each half exactly like the scalar ops, and
// a + (-b) == a - b, so the bytes do not change; what changes is the instruction count (28 instead of 56
// per 8-point transform).  Row pass ("horizontal"): a row's 8 values sit in 4 adjacent register pairs
// (p0,p1)(p2,p3)(p4,p5)(p6,p7); op_sel / neg modifiers pick the halves so that every butterfly stage is
// one packed op on two DIFFERENT quantities: (x07p,x16p), (x25p,x34p), (x07m,x61m), (x25m,x43m), (pp,qp),
// (pm,qm), (o0,o4), (o2,o6), (t1,t3), (t5,t7), (u1,u3), (u5,u7), (o1,o3), (o5,o7).  The outputs come out paired
// (0,4)(2,6)(1,3)(5,7) along u, identically for every row, so the column pass ("vertical") is the plain
// butterfly on 4 column pairs.  No register moves anywhere.
// =====================================================================================================
typedef float f2 __attribute__((ext_vector_type(2)));
struct XPkConsts
{
  f2 af, cd, be, nm; // (Ca,Cf) (Cc,Cd) (Cb,Ce) (Cn, magic)
};
#define PKA(d, a, b, mods) asm("v_pk_add_f32 %0, %1, %2 " mods : "=v"(d) : "v"(a), "v"(b))
#define PKM(d, a, k, mods) asm("v_pk_mul_f32 %0, %1, %2 " mods : "=v"(d) : "v"(a), "s"(k))
// halves of the constant operand (src1): LL = (lo,lo), HH = (hi,hi), LH = as stored, HL = swapped
#define K_LL "op_sel:[0,0] op_sel_hi:[1,0]"
#define K_HH "op_sel:[0,1] op_sel_hi:[1,1]"
#define K_LH "op_sel:[0,0] op_sel_hi:[1,1]"
#define K_HL "op_sel:[0,1] op_sel_hi:[1,0]"
#define X_CROSS "op_sel:[0,1] op_sel_hi:[1,0]" // lo = a.lo (+) b.hi, hi = a.hi (+) b.lo

// 8-point K_AVX transform of one row held in 4 pairs; outputs o04, o26, o13, o57 (already times Cn)
__device__ __forceinline__ void x_dct8_avx_h(const XPkConsts &K, f2 a01, f2 a23, f2 a45, f2 a67, f2 &o04, f2 &o26, f2 &o13, f2 &o57)
{
  f2 s1, s2, d, e, pqp, pqm, r, t, m1, m2, m3, m4, n1, n2, n3, n4, t13, t57, u13, u57;
  PKA(s1, a01, a67, X_CROSS);                                   // (p0+p7, p1+p6)
  PKA(s2, a23, a45, X_CROSS);                                   // (p2+p5, p3+p4)
  PKA(d, a01, a67, X_CROSS " neg_lo:[0,1] neg_hi:[1,0]");       // (p0-p7, p6-p1)
  PKA(e, a23, a45, X_CROSS " neg_lo:[0,1] neg_hi:[1,0]");       // (p2-p5, p4-p3)
  PKA(pqp, s1, s2, X_CROSS);                                    // (x07p+x34p, x16p+x25p)
  PKA(pqm, s1, s2, X_CROSS " neg_lo:[0,1] neg_hi:[0,1]");       // (x07p-x34p, x16p-x25p)
  PKA(o04, pqp, pqp, "op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]"); // (pp+qp, pp-qp)
  PKM(r, pqm, K.be, K_LL);                                      // (Cb pm, Cb qm)
  PKM(t, pqm, K.be, K_HH);                                      // (Ce pm, Ce qm)
  PKA(o26, r, t, X_CROSS " neg_hi:[1,0]");                      // (Cb pm + Ce qm, Ce pm - Cb qm)
  PKM(m1, d, K.af, K_LH);                                       // (Ca x07m, Cf x61m)
  PKM(m2, d, K.cd, K_LL);                                       // (Cc x07m, Cc x61m)
  PKM(m3, d, K.cd, K_HH);                                       // (Cd x07m, Cd x61m)
  PKM(m4, d, K.af, K_HL);                                       // (Cf x07m, Ca x61m)
  PKA(t13, m1, m2, X_CROSS " neg_lo:[0,1]");                    // (Ca x07m - Cc x61m, Cf x61m + Cc x07m)
  PKA(t57, m3, m4, X_CROSS);                                    // (Cd x07m + Ca x61m, Cd x61m + Cf x07m)
  PKM(n1, e, K.cd, K_HH);                                       // (Cd x25m, Cd x43m)
  PKM(n2, e, K.af, K_LH);                                       // (Ca x25m, Cf x43m)
  PKM(n3, e, K.af, K_HL);                                       // (Cf x25m, Ca x43m)
  PKM(n4, e, K.cd, K_LL);                                       // (Cc x25m, Cc x43m)
  PKA(u13, n1, n2, X_CROSS " neg_lo:[0,1]");                    // (Cd x25m - Cf x43m, Cd x43m + Ca x25m)
  PKA(u57, n3, n4, X_CROSS " neg_lo:[0,1]");                    // (Cf x25m - Cc x43m, Ca x43m + Cc x25m)
  PKA(o13, t13, u13, "neg_hi:[0,1]");                           // (t1 + u1, t3 - u3)
  PKA(o57, t57, u57, "");                                       // (t5 + u5, t7 + u7)
  PKM(o04, o04, K.nm, K_LL);
  PKM(o26, o26, K.nm, K_LL);
  PKM(o13, o13, K.nm, K_LL);
  PKM(o57, o57, K.nm, K_LL);
}

// the same transform down a column PAIR: p[r] = (B[r][u1], B[r][u2]), plain packed butterfly, in place
__device__ __forceinline__ void x_dct8_avx_v(const XPkConsts &K, f2 (&p)[8])
{
  f2 x07p, x16p, x25p, x34p, x07m, x61m, x25m, x43m, pp, pm, qp, qm, o0, o4, a, b, o2, o6;
  PKA(x07p, p[0], p[7], ""); PKA(x16p, p[1], p[6], ""); PKA(x25p, p[2], p[5], ""); PKA(x34p, p[3], p[4], "");
  PKA(x07m, p[0], p[7], "neg_lo:[0,1] neg_hi:[0,1]"); PKA(x61m, p[6], p[1], "neg_lo:[0,1] neg_hi:[0,1]");
  PKA(x25m, p[2], p[5], "neg_lo:[0,1] neg_hi:[0,1]"); PKA(x43m, p[4], p[3], "neg_lo:[0,1] neg_hi:[0,1]");
  PKA(pp, x07p, x34p, ""); PKA(pm, x07p, x34p, "neg_lo:[0,1] neg_hi:[0,1]");
  PKA(qp, x16p, x25p, ""); PKA(qm, x16p, x25p, "neg_lo:[0,1] neg_hi:[0,1]");
  PKA(o0, pp, qp, ""); PKA(o4, pp, qp, "neg_lo:[0,1] neg_hi:[0,1]");
  PKM(a, pm, K.be, K_LL); PKM(b, qm, K.be, K_HH); PKA(o2, a, b, "");                              // Cb pm + Ce qm
  PKM(a, pm, K.be, K_HH); PKM(b, qm, K.be, K_LL); PKA(o6, a, b, "neg_lo:[0,1] neg_hi:[0,1]");   // Ce pm - Cb qm
  f2 t1, t3, t5, t7, u1, u3, u5, u7, c, dd;
  PKM(a, x07m, K.af, K_LL); PKM(b, x61m, K.cd, K_LL); PKA(t1, a, b, "neg_lo:[0,1] neg_hi:[0,1]");  // Ca x07m - Cc x61m
  PKM(a, x07m, K.cd, K_LL); PKM(b, x61m, K.af, K_HH); PKA(t3, a, b, "");                          // Cc x07m + Cf x61m
  PKM(a, x07m, K.cd, K_HH); PKM(b, x61m, K.af, K_LL); PKA(t5, a, b, "");                          // Cd x07m + Ca x61m
  PKM(a, x07m, K.af, K_HH); PKM(b, x61m, K.cd, K_HH); PKA(t7, a, b, "");                          // Cf x07m + Cd x61m
  PKM(c, x25m, K.cd, K_HH); PKM(dd, x43m, K.af, K_HH); PKA(u1, c, dd, "neg_lo:[0,1] neg_hi:[0,1]"); // Cd x25m - Cf x43m
  PKM(c, x25m, K.af, K_LL); PKM(dd, x43m, K.cd, K_HH); PKA(u3, c, dd, "");                          // Ca x25m + Cd x43m
  PKM(c, x25m, K.af, K_HH); PKM(dd, x43m, K.cd, K_LL); PKA(u5, c, dd, "neg_lo:[0,1] neg_hi:[0,1]"); // Cf x25m - Cc x43m
  PKM(c, x25m, K.cd, K_LL); PKM(dd, x43m, K.af, K_LL); PKA(u7, c, dd, "");                          // Cc x25m + Ca x43m
  f2 o1, o3, o5, o7;
  PKA(o1, t1, u1, ""); PKA(o3, t3, u3, "neg_lo:[0,1] neg_hi:[0,1]"); PKA(o5, t5, u5, ""); PKA(o7, t7, u7, "");
  PKM(p[0], o0, K.nm, K_LL); PKM(p[1], o1, K.nm, K_LL); PKM(p[2], o2, K.nm, K_LL); PKM(p[3], o3, K.nm, K_LL);
  PKM(p[4], o4, K.nm, K_LL); PKM(p[5], o5, K.nm, K_LL); PKM(p[6], o6, K.nm, K_LL); PKM(p[7], o7, K.nm, K_LL);
}

// quantiser multipliers in pair order: pairs[v*4 + j] = (-q[v*8 + ua[j]], -q[v*8 + ub[j]]) with (ua,ub) = (0,4)(2,6)(1,3)(5,7)
struct PkQuant
{
  f2 nq[32];
};
__device__ __constant__ const int xPairA[4] = {0, 2, 1, 5};
__device__ __constant__ const int xPairB[4] = {4, 6, 3, 7};

// rows (raw bytes) -> 64 words whose low byte is ~(reference byte), natural index v*8+u
__device__ __forceinline__ void transform_quant_pk(const XPkConsts &K, const PkQuant &Q, const uint2 (&rows)[8], uint32_t (&out)[64])
{
  f2 col[4][8]; // col[j][r] = (B[r][ua_j], B[r][ub_j]) after the row pass
#pragma unroll
  for (int r = 0; r < 8; r++)
  {
    f2 a01 = {ubyte_to_float<0>(rows[r].x), ubyte_to_float<1>(rows[r].x)};
    f2 a23 = {ubyte_to_float<2>(rows[r].x), ubyte_to_float<3>(rows[r].x)};
    f2 a45 = {ubyte_to_float<0>(rows[r].y), ubyte_to_float<1>(rows[r].y)};
    f2 a67 = {ubyte_to_float<2>(rows[r].y), ubyte_to_float<3>(rows[r].y)};
    x_dct8_avx_h(K, a01, a23, a45, a67, col[0][r], col[1][r], col[2][r], col[3][r]);
  }
  constexpr int ua[4] = {0, 2, 1, 5}, ub[4] = {4, 6, 3, 7};
#pragma unroll
  for (int j = 0; j < 4; j++)
  {
    x_dct8_avx_v(K, col[j]);
#pragma unroll
    for (int v = 0; v < 8; v++)
    {
      f2 m, t;
      PKM(m, col[j][v], Q.nq[v * 4 + j], K_LH);
      m.x = __builtin_amdgcn_fmed3f(m.x, -128.0f, 127.0f);
      m.y = __builtin_amdgcn_fmed3f(m.y, -128.0f, 127.0f);
      PKA(t, m, K.nm, "op_sel:[0,1] op_sel_hi:[1,1]"); // + (magic, magic)
      out[v * 8 + ua[j]] = __float_as_uint(t.x);
      out[v * 8 + ub[j]] = __float_as_uint(t.y);
    }
  }
}

template <int MINW>
__global__ __launch_bounds__(256, MINW) void v_pk(U8Args a, XPkConsts K, PkQuant Q)
{
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t row = t / a.bpr, bx = t - row * a.bpr;
  const uint8_t *src = a.from + (size_t)(a.by0 + row) * 8 * a.pitch + (size_t)bx * 8;
  __shared__ __attribute__((aligned(16))) uint8_t lds[4][64 * kQ32RowStride];
  uint2 rows[8];
  load_block_rows(src, a.pitch, rows);
  uint32_t q[64];
  transform_quant_pk(K, Q, rows, q);
  reorder_store<true>(lds[threadIdx.x >> 6], q, lane, lane, a.to + ((size_t)a.by0 * a.bpr + (t - lane)) * 64);
}

template <int MINW, int NPASS>
__global__ __launch_bounds__(256, MINW) void v_pk_dma(U8Args a, XPkConsts K, PkQuant Q, uint32_t ntiles)
{
  __shared__ __attribute__((aligned(16))) uint8_t stage[4][(64 / NPASS) * kQ32RowStride];
  __shared__ __attribute__((aligned(16))) uint8_t inbuf[4][8 * 512];
  const uint32_t lane = threadIdx.x & 63, half = lane >> 5, l32 = lane & 31;
  const uint32_t w = threadIdx.x >> 6;
  uint8_t *wl = stage[w];
  uint8_t *in = inbuf[w];
  const uint32_t total_waves = gridDim.x * 4;
  uint32_t tile = blockIdx.x * 4 + w;
  if (tile >= ntiles)
    return;
  auto issue = [&](uint32_t tl) {
    const uint32_t t0 = tl * 64;
    const uint32_t row = t0 / a.bpr, bx0 = t0 - row * a.bpr;
    const uint8_t *src16 = a.from + ((size_t)(a.by0 + row) * 8 + half) * a.pitch + (size_t)bx0 * 8 + l32 * 16;
#pragma unroll
    for (int k = 0; k < 4; k++)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src16 + (size_t)(2 * k) * a.pitch),
                                       (__attribute__((address_space(3))) void *)(in + k * 1024), 16, 0, 0);
  };
  issue(tile);
  __builtin_amdgcn_s_waitcnt(0x0F70);
  for (;;)
  {
    uint2 rows[8];
#pragma unroll
    for (int r = 0; r < 8; r++)
      rows[r] = *reinterpret_cast<const uint2 *>(in + r * 512 + lane * 8);
    __builtin_amdgcn_s_waitcnt(0xC07F); // lgkmcnt(0): the input buffer is free
    const uint32_t next = tile + total_waves;
    const bool more = next < ntiles;
    if (more)
      issue(next);
    uint32_t q[64];
    transform_quant_pk(K, Q, rows, q);
    reorder_store_passes<NPASS>(wl, q, lane, lane, a.to + ((size_t)a.by0 * a.bpr + (size_t)tile * 64) * 64);
    if (!more)
      break;
    tile = next;
  }
}


// ---- v_pk_dma2: scalar (SGPR) tile bookkeeping, rows read from LDS just in time during the row pass, the next
// tile's DMA issued AFTER the row pass (it still has the column pass + quantise + reorder to land)
template <int MINW, int NPASS>
__global__ __launch_bounds__(256, MINW) void v_pk_dma2(U8Args a, XPkConsts K, PkQuant Q, uint32_t ntiles)
{
  __shared__ __attribute__((aligned(16))) uint8_t stage[4][(64 / NPASS) * kQ32RowStride];
  __shared__ __attribute__((aligned(16))) uint8_t inbuf[4][8 * 512];
  const uint32_t lane = threadIdx.x & 63, half = lane >> 5, l32 = lane & 31;
  const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  uint8_t *wl = stage[w];
  uint8_t *in = inbuf[w];
  const uint32_t total_waves = gridDim.x * 4;
  uint32_t tile = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + w);
  if (tile >= ntiles)
    return;
  const uint32_t lane_in = half * (uint32_t)a.pitch + l32 * 16; // per-lane part of the source address (plane < 4 GiB)
  auto issue = [&](uint32_t tl) {
    const uint32_t t0 = tl * 64;
    const uint32_t row = t0 / a.bpr, bx0 = t0 - row * a.bpr;
    const uint8_t *base = a.from + (size_t)(a.by0 + row) * 8 * a.pitch + (size_t)bx0 * 8; // wave-uniform
#pragma unroll
    for (int k = 0; k < 4; k++)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + (size_t)(2 * k) * a.pitch + lane_in),
                                       (__attribute__((address_space(3))) void *)(in + k * 1024), 16, 0, 0);
  };
  issue(tile);
  __builtin_amdgcn_s_waitcnt(0x0F70);
  for (;;)
  {
    f2 col[4][8];
#pragma unroll
    for (int r = 0; r < 8; r++)
    {
      const uint2 rw = *reinterpret_cast<const uint2 *>(in + r * 512 + lane * 8);
      f2 a01 = {ubyte_to_float<0>(rw.x), ubyte_to_float<1>(rw.x)};
      f2 a23 = {ubyte_to_float<2>(rw.x), ubyte_to_float<3>(rw.x)};
      f2 a45 = {ubyte_to_float<0>(rw.y), ubyte_to_float<1>(rw.y)};
      f2 a67 = {ubyte_to_float<2>(rw.y), ubyte_to_float<3>(rw.y)};
      x_dct8_avx_h(K, a01, a23, a45, a67, col[0][r], col[1][r], col[2][r], col[3][r]);
    }
    const uint32_t next = tile + total_waves;
    const bool more = next < ntiles;
    if (more)
      issue(next); // every LDS read of the input buffer has been consumed by the row pass above
    uint32_t q[64];
    constexpr int ua[4] = {0, 2, 1, 5}, ub[4] = {4, 6, 3, 7};
#pragma unroll
    for (int j = 0; j < 4; j++)
    {
      x_dct8_avx_v(K, col[j]);
#pragma unroll
      for (int v = 0; v < 8; v++)
      {
        f2 m, t;
        PKM(m, col[j][v], Q.nq[v * 4 + j], K_LH);
        m.x = __builtin_amdgcn_fmed3f(m.x, -128.0f, 127.0f);
        m.y = __builtin_amdgcn_fmed3f(m.y, -128.0f, 127.0f);
        PKA(t, m, K.nm, "op_sel:[0,1] op_sel_hi:[1,1]");
        q[v * 8 + ua[j]] = __float_as_uint(t.x);
        q[v * 8 + ub[j]] = __float_as_uint(t.y);
      }
    }
    reorder_store_passes<NPASS>(wl, q, lane, lane, a.to + ((size_t)a.by0 * a.bpr + (size_t)tile * 64) * 64);
    if (!more)
      break;
    tile = next;
  }
}


// ---- v_pk_dma3: v_pk_dma2 with scheduling barriers after every line of the row pass and every column pair, so that
// the compiler cannot hoist the LDS reads / spread the live ranges (register pressure); scalar (SGPR) tile bookkeeping, rows read from LDS just in time during the row pass, the next
// tile's DMA issued AFTER the row pass (it still has the column pass + quantise + reorder to land)
template <int MINW, int NPASS>
__global__ __launch_bounds__(256, MINW) void v_pk_dma3(U8Args a, XPkConsts K, PkQuant Q, uint32_t ntiles)
{
  __shared__ __attribute__((aligned(16))) uint8_t stage[4][(64 / NPASS) * kQ32RowStride];
  __shared__ __attribute__((aligned(16))) uint8_t inbuf[4][8 * 512];
  const uint32_t lane = threadIdx.x & 63, half = lane >> 5, l32 = lane & 31;
  const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  uint8_t *wl = stage[w];
  uint8_t *in = inbuf[w];
  const uint32_t total_waves = gridDim.x * 4;
  uint32_t tile = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + w);
  if (tile >= ntiles)
    return;
  const uint32_t lane_in = half * (uint32_t)a.pitch + l32 * 16; // per-lane part of the source address (plane < 4 GiB)
  auto issue = [&](uint32_t tl) {
    const uint32_t t0 = tl * 64;
    const uint32_t row = t0 / a.bpr, bx0 = t0 - row * a.bpr;
    const uint8_t *base = a.from + (size_t)(a.by0 + row) * 8 * a.pitch + (size_t)bx0 * 8; // wave-uniform
#pragma unroll
    for (int k = 0; k < 4; k++)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + (size_t)(2 * k) * a.pitch + lane_in),
                                       (__attribute__((address_space(3))) void *)(in + k * 1024), 16, 0, 0);
  };
  issue(tile);
  __builtin_amdgcn_s_waitcnt(0x0F70);
  for (;;)
  {
    f2 col[4][8];
#pragma unroll
    for (int r = 0; r < 8; r++)
    {
      const uint2 rw = *reinterpret_cast<const uint2 *>(in + r * 512 + lane * 8);
      f2 a01 = {ubyte_to_float<0>(rw.x), ubyte_to_float<1>(rw.x)};
      f2 a23 = {ubyte_to_float<2>(rw.x), ubyte_to_float<3>(rw.x)};
      f2 a45 = {ubyte_to_float<0>(rw.y), ubyte_to_float<1>(rw.y)};
      f2 a67 = {ubyte_to_float<2>(rw.y), ubyte_to_float<3>(rw.y)};
      x_dct8_avx_h(K, a01, a23, a45, a67, col[0][r], col[1][r], col[2][r], col[3][r]);
      __builtin_amdgcn_sched_barrier(0);
    }
    const uint32_t next = tile + total_waves;
    const bool more = next < ntiles;
    if (more)
      issue(next); // every LDS read of the input buffer has been consumed by the row pass above
    uint32_t q[64];
    constexpr int ua[4] = {0, 2, 1, 5}, ub[4] = {4, 6, 3, 7};
#pragma unroll
    for (int j = 0; j < 4; j++)
    {
      x_dct8_avx_v(K, col[j]);
#pragma unroll
      for (int v = 0; v < 8; v++)
      {
        f2 m, t;
        PKM(m, col[j][v], Q.nq[v * 4 + j], K_LH);
        m.x = __builtin_amdgcn_fmed3f(m.x, -128.0f, 127.0f);
        m.y = __builtin_amdgcn_fmed3f(m.y, -128.0f, 127.0f);
        PKA(t, m, K.nm, "op_sel:[0,1] op_sel_hi:[1,1]");
        q[v * 8 + ua[j]] = __float_as_uint(t.x);
        q[v * 8 + ub[j]] = __float_as_uint(t.y);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    reorder_store_passes<NPASS>(wl, q, lane, lane, a.to + ((size_t)a.by0 * a.bpr + (size_t)tile * 64) * 64);
    if (!more)
      break;
    tile = next;
  }
}


// ---- VALU floor of the packed transform: REPS x (load + transform + quantise) on the same tile (L2 hits after the
// first), results folded into 16 bytes; (t(REPS=3) - t(REPS=1)) / 2 is one compute pass with HBM out of the picture
template <int REPS, int MINW>
__global__ __launch_bounds__(256, MINW) void v_pk_reps(U8Args a, XPkConsts K, PkQuant Q)
{
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  const uint32_t row = t / a.bpr, bx = t - row * a.bpr;
  const uint8_t *src = a.from + (size_t)(a.by0 + row) * 8 * a.pitch + (size_t)bx * 8;
  uint4 acc = make_uint4(0, 0, 0, 0);
#pragma unroll 1
  for (int i = 0; i < REPS; i++)
  {
    uint2 rows[8];
    load_block_rows(src + (acc.x & 8), a.pitch, rows); // address depends on the previous repetition
    uint32_t q[64];
    transform_quant_pk(K, Q, rows, q);
#pragma unroll
    for (int c = 0; c < 64; c += 4)
    {
      acc.x ^= q[c]; acc.y ^= q[c + 1]; acc.z ^= q[c + 2]; acc.w ^= q[c + 3];
    }
  }
  *reinterpret_cast<uint4 *>(a.to + (size_t)t * 64) = acc;
}


// ---- static priorities by wave slot: the waves a SIMD hosts start together and, arbitrated round-robin, finish
// together; distinct priorities serialise them (the highest finishes first, its successor starts loading early)
__device__ __forceinline__ void prio_by_slot(int mode)
{
  const uint32_t slot = __builtin_amdgcn_s_getreg(0x1804) & 15; // HW_REG_HW_ID[3:0]: wave slot within the SIMD
  const uint32_t k = mode == 0 ? (slot & 3) : (mode == 1 ? (slot & 1) : ((slot >> 1) & 3));
  if (k == 1) __builtin_amdgcn_s_setprio(1);
  else if (k == 2) __builtin_amdgcn_s_setprio(2);
  else if (k == 3) __builtin_amdgcn_s_setprio(3);
}
template <int MINW, int MODE>
__global__ __launch_bounds__(256, MINW) void v_pk_prio(U8Args a, XPkConsts K, PkQuant Q)
{
  prio_by_slot(MODE);
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t row = t / a.bpr, bx = t - row * a.bpr;
  const uint8_t *src = a.from + (size_t)(a.by0 + row) * 8 * a.pitch + (size_t)bx * 8;
  __shared__ __attribute__((aligned(16))) uint8_t lds[4][64 * kQ32RowStride];
  uint2 rows[8];
  load_block_rows(src, a.pitch, rows);
  uint32_t q[64];
  transform_quant_pk(K, Q, rows, q);
  reorder_store<true>(lds[threadIdx.x >> 6], q, lane, lane, a.to + ((size_t)a.by0 * a.bpr + (t - lane)) * 64);
}

// ---- memory-only shapes: what do 8 B/lane vs 16 B/lane non-temporal row loads cost with no arithmetic?
template <bool WIDE>
__global__ __launch_bounds__(256) void v_memonly(U8Args a)
{
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  const uint32_t lane = threadIdx.x & 63, half = lane >> 5, l32 = lane & 31;
  const uint32_t t0 = t - lane;
  const uint32_t row = t0 / a.bpr, bx0 = t0 - row * a.bpr;
  u32x4_t o[4];
  if (WIDE)
  {
    const uint8_t *src16 = a.from + ((size_t)(a.by0 + row) * 8 + half) * a.pitch + (size_t)bx0 * 8 + l32 * 16;
    load_wide(src16, a.pitch, o);
  }
  else
  {
    const uint8_t *src = a.from + (size_t)(a.by0 + row) * 8 * a.pitch + (size_t)(bx0 + lane) * 8;
    uint2 rows[8];
    load_block_rows(src, a.pitch, rows);
#pragma unroll
    for (int k = 0; k < 4; k++)
      o[k] = u32x4_t{rows[2 * k].x, rows[2 * k].y, rows[2 * k + 1].x, rows[2 * k + 1].y};
  }
  uint8_t *outw = a.to + ((size_t)a.by0 * a.bpr + t0) * 64;
#pragma unroll
  for (int k = 0; k < 4; k++)
    __builtin_nontemporal_store(o[k], reinterpret_cast<u32x4_t *>(outw + k * 1024 + lane * 16));
}

int main(int argc, char **argv)
{
  const bool pmc = argc > 1 && !strcmp(argv[1], "pmc"); // few launches per variant, for rocprofv3 --pmc passes
  const size_t W = 8192, H = 8192, bytes = W * H;
  const int NS = 4;
  std::vector<uint8_t *> A(NS), B(NS);
  std::vector<uint8_t> host(W * H);
  for (size_t i = 0; i < W * H; i++) host[i] = (uint8_t)((i * 2654435761u) >> 24);
  for (int i = 0; i < NS; i++)
  {
    if (hipMalloc(&A[i], bytes + 64) != hipSuccess || hipMalloc(&B[i], bytes) != hipSuccess) { puts("alloc failed"); return 1; }
    hipMemcpy(A[i], host.data(), bytes, hipMemcpyHostToDevice);
  }
  U8Args a;
  memset(&a, 0, sizeof(a));
  a.consts = DctConsts();
  for (int i = 0; i < 64; i++) a.qt.q[i] = 255.0f / ((0.1f + 0.01f * i) * 2000 * 0.95f);
  a.pitch = W; a.sizeX = W; a.out_strip = 8 * W; a.out_tight = 1; a.bpr = W / 8; a.by0 = 0; a.nblocks = (uint32_t)(W / 8 * H / 8);
  U8Args an = a;
  for (int i = 0; i < 64; i++) an.qt.q[i] = -a.qt.q[i];
  const float magicC = 12582912.0f + 128.0f;
  XPkConsts PK;
  PK.af = f2{a.consts.a, a.consts.f}; PK.cd = f2{a.consts.c, a.consts.d}; PK.be = f2{a.consts.b, a.consts.e}; PK.nm = f2{a.consts.n, magicC};
  PkQuant PQ;
  {
    const int ua[4] = {0, 2, 1, 5}, ub[4] = {4, 6, 3, 7};
    for (int v = 0; v < 8; v++)
      for (int j = 0; j < 4; j++)
        PQ.nq[v * 4 + j] = f2{-a.qt.q[v * 8 + ua[j]], -a.qt.q[v * 8 + ub[j]]};
  }
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  struct V { const char *name; std::function<void(int)> f; std::vector<float> t; bool check; };
  std::vector<V> vs;
  auto args = [&](const U8Args &base, int s) { U8Args x = base; x.from = A[s]; x.to = B[s]; return x; };
  const uint32_t ntiles = a.nblocks / 64, nwg = a.nblocks / 256;
  U8Args ap = a; // what mdct_api.hip hands the product kernel: pair-ordered negated multipliers + packed constants
  for (int v = 0; v < 8; v++)
    for (int j = 0; j < 4; j++)
    {
      ap.qt.q[(v * 4 + j) * 2] = -a.qt.q[v * 8 + kPairA[j]];
      ap.qt.q[(v * 4 + j) * 2 + 1] = -a.qt.q[v * 8 + kPairB[j]];
    }
  ap.pk = PkConstsArg{{a.consts.a, a.consts.f}, {a.consts.c, a.consts.d}, {a.consts.b, a.consts.e}, {a.consts.n, magicC}};
  vs.push_back({"product q32", [&](int s) { launch_fwd_quant_u8(args(ap, s), MDCT_LAYOUT_Q32, MDCT_PROFILE_REF_AVX, false, 0); }, {}, false});
  vs.push_back({"mem only 8B/lane rows", [&](int s) { hipLaunchKernelGGL((v_memonly<false>), dim3(nwg), dim3(256), 0, 0, args(a, s)); }, {}, false});
  vs.push_back({"mem only 16B/lane rows", [&](int s) { hipLaunchKernelGGL((v_memonly<true>), dim3(nwg), dim3(256), 0, 0, args(a, s)); }, {}, false});
  vs.push_back({"not 6w", [&](int s) { hipLaunchKernelGGL((v_not<6>), dim3(nwg), dim3(256), 0, 0, args(an, s), magicC); }, {}, true});
  vs.push_back({"not 5w", [&](int s) { hipLaunchKernelGGL((v_not<5>), dim3(nwg), dim3(256), 0, 0, args(an, s), magicC); }, {}, true});
  vs.push_back({"pipe 4w grid 1024", [&](int s) { hipLaunchKernelGGL((v_pipe<4>), dim3(1024), dim3(256), 0, 0, args(an, s), magicC, ntiles); }, {}, true});
  vs.push_back({"dma 4w grid 1024", [&](int s) { hipLaunchKernelGGL((v_dma<4>), dim3(1024), dim3(256), 0, 0, args(an, s), magicC, ntiles); }, {}, true});
  vs.push_back({"dma2 4w 1pass grid 1024", [&](int s) { hipLaunchKernelGGL((v_dma2<4, 1>), dim3(1024), dim3(256), 0, 0, args(an, s), magicC, ntiles); }, {}, true});
  vs.push_back({"dma2 5w 2pass grid 1280", [&](int s) { hipLaunchKernelGGL((v_dma2<5, 2>), dim3(1280), dim3(256), 0, 0, args(an, s), magicC, ntiles); }, {}, true});
  vs.push_back({"dma2 6w 4pass grid 1536", [&](int s) { hipLaunchKernelGGL((v_dma2<6, 4>), dim3(1536), dim3(256), 0, 0, args(an, s), magicC, ntiles); }, {}, true});
  vs.push_back({"not 6w 4pass (no dma)", [&](int s) { hipLaunchKernelGGL((v_not_passes<6, 4>), dim3(nwg), dim3(256), 0, 0, args(an, s), magicC); }, {}, true});
  vs.push_back({"pk 6w", [&](int s) { hipLaunchKernelGGL((v_pk<6>), dim3(nwg), dim3(256), 0, 0, args(a, s), PK, PQ); }, {}, true});
  vs.push_back({"pk 5w", [&](int s) { hipLaunchKernelGGL((v_pk<5>), dim3(nwg), dim3(256), 0, 0, args(a, s), PK, PQ); }, {}, true});
  vs.push_back({"pk 4w", [&](int s) { hipLaunchKernelGGL((v_pk<4>), dim3(nwg), dim3(256), 0, 0, args(a, s), PK, PQ); }, {}, true});
  vs.push_back({"pk dma 4w 1pass grid 1024", [&](int s) { hipLaunchKernelGGL((v_pk_dma<4, 1>), dim3(1024), dim3(256), 0, 0, args(a, s), PK, PQ, ntiles); }, {}, true});
  vs.push_back({"pk dma 5w 2pass grid 1280", [&](int s) { hipLaunchKernelGGL((v_pk_dma<5, 2>), dim3(1280), dim3(256), 0, 0, args(a, s), PK, PQ, ntiles); }, {}, true});
  vs.push_back({"pk dma 6w 2pass grid 1536", [&](int s) { hipLaunchKernelGGL((v_pk_dma<6, 2>), dim3(1536), dim3(256), 0, 0, args(a, s), PK, PQ, ntiles); }, {}, true});
  vs.push_back({"pk dma 6w 4pass grid 1536", [&](int s) { hipLaunchKernelGGL((v_pk_dma<6, 4>), dim3(1536), dim3(256), 0, 0, args(a, s), PK, PQ, ntiles); }, {}, true});
  vs.push_back({"pk dma2 4w 1pass grid 1024", [&](int s) { hipLaunchKernelGGL((v_pk_dma2<4, 1>), dim3(1024), dim3(256), 0, 0, args(a, s), PK, PQ, ntiles); }, {}, true});
  vs.push_back({"pk dma2 5w 2pass grid 1280", [&](int s) { hipLaunchKernelGGL((v_pk_dma2<5, 2>), dim3(1280), dim3(256), 0, 0, args(a, s), PK, PQ, ntiles); }, {}, true});
  vs.push_back({"pk dma2 6w 2pass grid 1536", [&](int s) { hipLaunchKernelGGL((v_pk_dma2<6, 2>), dim3(1536), dim3(256), 0, 0, args(a, s), PK, PQ, ntiles); }, {}, true});
  vs.push_back({"pk dma2 6w 4pass grid 1536", [&](int s) { hipLaunchKernelGGL((v_pk_dma2<6, 4>), dim3(1536), dim3(256), 0, 0, args(a, s), PK, PQ, ntiles); }, {}, true});
  vs.push_back({"pk reps x1 6w", [&](int s) { hipLaunchKernelGGL((v_pk_reps<1, 6>), dim3(nwg), dim3(256), 0, 0, args(a, s), PK, PQ); }, {}, false});
  vs.push_back({"pk reps x3 6w", [&](int s) { hipLaunchKernelGGL((v_pk_reps<3, 6>), dim3(nwg), dim3(256), 0, 0, args(a, s), PK, PQ); }, {}, false});
  vs.push_back({"pk reps x5 6w", [&](int s) { hipLaunchKernelGGL((v_pk_reps<5, 6>), dim3(nwg), dim3(256), 0, 0, args(a, s), PK, PQ); }, {}, false});
  vs.push_back({"pk reps x1 5w", [&](int s) { hipLaunchKernelGGL((v_pk_reps<1, 5>), dim3(nwg), dim3(256), 0, 0, args(a, s), PK, PQ); }, {}, false});
  vs.push_back({"pk reps x3 5w", [&](int s) { hipLaunchKernelGGL((v_pk_reps<3, 5>), dim3(nwg), dim3(256), 0, 0, args(a, s), PK, PQ); }, {}, false});
  vs.push_back({"pk reps x1 4w", [&](int s) { hipLaunchKernelGGL((v_pk_reps<1, 4>), dim3(nwg), dim3(256), 0, 0, args(a, s), PK, PQ); }, {}, false});
  vs.push_back({"pk reps x3 4w", [&](int s) { hipLaunchKernelGGL((v_pk_reps<3, 4>), dim3(nwg), dim3(256), 0, 0, args(a, s), PK, PQ); }, {}, false});
  vs.push_back({"pk 6w prio slot&3", [&](int s) { hipLaunchKernelGGL((v_pk_prio<6, 0>), dim3(nwg), dim3(256), 0, 0, args(a, s), PK, PQ); }, {}, true});
  vs.push_back({"pk 6w prio slot&1", [&](int s) { hipLaunchKernelGGL((v_pk_prio<6, 1>), dim3(nwg), dim3(256), 0, 0, args(a, s), PK, PQ); }, {}, true});
  vs.push_back({"pk 6w prio (slot>>1)&3", [&](int s) { hipLaunchKernelGGL((v_pk_prio<6, 2>), dim3(nwg), dim3(256), 0, 0, args(a, s), PK, PQ); }, {}, true});
  vs.push_back({"pk 5w prio slot&3", [&](int s) { hipLaunchKernelGGL((v_pk_prio<5, 0>), dim3(nwg), dim3(256), 0, 0, args(a, s), PK, PQ); }, {}, true});
  vs.push_back({"pk dma3 5w 2pass grid 1280", [&](int s) { hipLaunchKernelGGL((v_pk_dma3<5, 2>), dim3(1280), dim3(256), 0, 0, args(a, s), PK, PQ, ntiles); }, {}, true});
  vs.push_back({"pk dma3 6w 2pass grid 1536", [&](int s) { hipLaunchKernelGGL((v_pk_dma3<6, 2>), dim3(1536), dim3(256), 0, 0, args(a, s), PK, PQ, ntiles); }, {}, true});
  vs.push_back({"pk dma3 4w 1pass grid 1024", [&](int s) { hipLaunchKernelGGL((v_pk_dma3<4, 1>), dim3(1024), dim3(256), 0, 0, args(a, s), PK, PQ, ntiles); }, {}, true});
  // correctness of every variant against the product kernel's bytes
  {
    std::vector<uint8_t> ref(bytes), got(bytes);
    vs[0].f(0); hipMemcpy(ref.data(), B[0], bytes, hipMemcpyDeviceToHost);
    for (auto &v : vs)
    {
      if (!v.check) continue;
      hipMemset(B[0], 0x55, bytes);
      v.f(0);
      if (hipMemcpy(got.data(), B[0], bytes, hipMemcpyDeviceToHost) != hipSuccess) { printf("!! %s: launch failed\n", v.name); return 1; }
      size_t bad = 0;
      for (size_t i = 0; i < bytes; i++) bad += got[i] != ref[i];
      printf("%-28s %s (%zu mismatching bytes)\n", v.name, bad ? "!! MISMATCH" : "bit-exact", bad);
    }
    fflush(stdout);
  }
  if (pmc)
  {
    for (auto &v : vs) for (int i = 0; i < 8; i++) v.f(i % NS);
    hipDeviceSynchronize();
    return 0;
  }
  for (auto &v : vs) for (int i = 0; i < 300; i++) v.f(i % NS);
  hipDeviceSynchronize();
  for (int round = 0; round < 7; round++)
    for (auto &v : vs)
    {
      for (int i = 0; i < 40; i++) v.f(i % NS); // re-enter each variant's own steady state
      hipEventRecord(e0, 0);
      for (int i = 0; i < 40; i++) v.f(i % NS);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      v.t.push_back(ms / 40);
    }
  for (auto &v : vs)
  {
    std::sort(v.t.begin(), v.t.end());
    printf("%-28s median %7.2f us  min %7.2f us\n", v.name, v.t[v.t.size() / 2] * 1e3, v.t[0] * 1e3);
  }
  return 0;
}
