// exp_u8_r3.hip -- round-3 diagnostics and A/B harness for the u8 kernels (q32 first).
//   timeline   the product q32 body with s_memrealtime / s_memtime stamps at the phase boundaries of every wave
//              (0 entry, 1 rows arrived, 2 transform done + bytes staged in LDS, 3 read back + stores issued, 4 stores acknowledged) and the
//              wave's placement (HW_ID, XCC_ID) -> gpurun_out/q32_timeline.bin, analysed by tools/q32_timeline.py.
//              A diagnostic build: the stamped kernel is never the product.
//   ab         variants against the product kernel's bytes, interleaved rounds in one process
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -std=c++17 -Iinclude -Isimd_dct_amd/csrc tools/exp_u8_r3.hip -o tools/exp_u8_r3
#include "../simd_dct_amd/csrc/mdct_kernels.hip"
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <functional>
#include <vector>
using namespace mdct;

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

#define GETREG(id) __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | (id))
struct Stamp
{
  uint32_t hw_id, xcc_id, wave_t0, pad;
  uint32_t rt[6]; // s_memrealtime (100 MHz, chip-wide), low 32 bits
  uint32_t ck[6]; // s_memtime (shader clock of the XCD), low 32 bits
};
static_assert(sizeof(Stamp) == 64, "16 dwords per wave");

// the stamp leaves at once (lane 0, 8 bytes): the kernel's SGPRs are full of quantiser multipliers, held stamps would spill
#define STAMP(i)                                                                                       \
  do                                                                                                   \
  {                                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
    const uint32_t rt_ = (uint32_t)__builtin_amdgcn_s_memrealtime();                                   \
    const uint32_t ck_ = (uint32_t)__builtin_amdgcn_s_memtime();                                       \
    if ((threadIdx.x & 63) == 0)                                                                       \
    {                                                                                                  \
      Stamp *st_ = stamps + ((blockIdx.x * kWG + (threadIdx.x & ~63u)) >> 6);                          \
      st_->rt[i] = rt_;                                                                                \
      st_->ck[i] = ck_;                                                                                \
    }                                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
  } while (0)

// the product kernel's fast path (k_q32_avx<false,false>), stamped
template <int MINW>
__global__ __launch_bounds__(kWG, MINW) void v_timeline(U8Args a, Stamp *stamps)
{
  __shared__ __attribute__((aligned(16))) uint8_t lds[kWG / 64][64 * kQ32RowStride];
  STAMP(0);
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t wave_t0 = blockIdx.x * kWG + wave * 64;
  const uint32_t t = wave_t0 + lane;
  uint32_t q[64];
  {
    const uint32_t row = t / a.bpr;
    const uint32_t bx = t - row * a.bpr;
    uint2 rows[8];
    load_block_rows(a.from + (size_t)(a.by0 + row) * 8 * a.pitch + (size_t)bx * 8, a.pitch, rows);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP(1);
    encode_block_avx_pk<false>(reinterpret_cast<const PkConsts &>(a.pk), rows, a.qt, q);
  }
  uint8_t *wl = lds[wave];
#pragma unroll
  for (int c = 0; c < 64; c++)
    wl[c * kQ32RowStride + lane] = (uint8_t)q[c];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  uint8_t *outw = a.to + ((size_t)a.by0 * a.bpr + wave_t0) * 64;
  const uint32_t c2 = (lane & 31) * 2;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // transform done and all 64 byte writes landed (the compiler interleaves them, as in the product)
  STAMP(2);
#pragma unroll
  for (int k = 0; k < 4; k++)
  {
    const uint32_t g = 2 * k + (lane >> 5);
    const uint2 lo = *reinterpret_cast<const uint2 *>(wl + c2 * kQ32RowStride + g * 8);
    const uint2 hi = *reinterpret_cast<const uint2 *>(wl + (c2 + 1) * kQ32RowStride + g * 8);
    const u32x4_t v = ~u32x4_t{lo.x, lo.y, hi.x, hi.y};
    __builtin_nontemporal_store(v, reinterpret_cast<u32x4_t *>(outw + g * 512 + c2 * 8));
  }
  STAMP(3);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  STAMP(4);
  if (lane == 0)
  {
    Stamp *s = stamps + (wave_t0 >> 6);
    s->hw_id = GETREG(4);
    s->xcc_id = GETREG(20);
    s->wave_t0 = wave_t0;
    s->pad = 0;
  }
}

// ---------------------------------------------------------------------------------------
// Variant: stores issued as soon as their coefficient pairs exist (review item ii).  The column
// pairs are processed in the order j = 0, 2, 3, 1 so that after two of them the coefficients
// (v, 0) and (v, 1) of all v are complete (one quarter of the wave's 16-byte chunks), after the
// third (v, 4), (v, 5), after the fourth the rest.
// ---------------------------------------------------------------------------------------
template <int MINW>
__global__ __launch_bounds__(kWG, MINW) void v_early(U8Args a)
{
  __shared__ __attribute__((aligned(16))) uint8_t lds[kWG / 64][64 * kQ32RowStride];
  const PkConsts &K = reinterpret_cast<const PkConsts &>(a.pk);
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t wave_t0 = blockIdx.x * kWG + wave * 64;
  const uint32_t t = wave_t0 + lane;
  const uint32_t row = t / a.bpr;
  const uint32_t bx = t - row * a.bpr;
  uint2 rows[8];
  load_block_rows(a.from + (size_t)(a.by0 + row) * 8 * a.pitch + (size_t)bx * 8, a.pitch, rows);
  f32x2 col[4][8];
#pragma unroll
  for (int r = 0; r < 8; r++)
  {
    const f32x2 a01 = {ubyte_to_float<0>(rows[r].x), ubyte_to_float<1>(rows[r].x)};
    const f32x2 a23 = {ubyte_to_float<2>(rows[r].x), ubyte_to_float<3>(rows[r].x)};
    const f32x2 a45 = {ubyte_to_float<0>(rows[r].y), ubyte_to_float<1>(rows[r].y)};
    const f32x2 a67 = {ubyte_to_float<2>(rows[r].y), ubyte_to_float<3>(rows[r].y)};
    dct8_h<K_AVX>(K, a01, a23, a45, a67, col[0][r], col[1][r], col[2][r], col[3][r]);
  }
  uint8_t *wl = lds[wave];
  uint8_t *outw = a.to + ((size_t)a.by0 * a.bpr + wave_t0) * 64;
  // 8-byte pieces: lane l of store s handles coefficient c = ..., group g: [coef c][8 blocks of group g] = 8 contiguous output bytes at g*512 + c*8
  auto quant_stage = [&](auto j_) {
    constexpr int j = decltype(j_)::value;
    dct8_v<K_AVX>(K, col[j]);
#pragma unroll
    for (int v = 0; v < 8; v++)
    {
      const f32x2 qp = reinterpret_cast<const f32x2 *>(a.qt.q)[v * 4 + j];
      f32x2 m, tt;
      MDCT_PKM(m, col[j][v], qp, MDCT_K_LH);
      m.x = __builtin_amdgcn_fmed3f(m.x, -128.0f, 127.0f);
      m.y = __builtin_amdgcn_fmed3f(m.y, -128.0f, 127.0f);
      MDCT_PKA(tt, m, K.nm, MDCT_K_HH);
      wl[(v * 8 + kPairA[j]) * kQ32RowStride + lane] = (uint8_t)__float_as_uint(tt.x);
      wl[(v * 8 + kPairB[j]) * kQ32RowStride + lane] = (uint8_t)__float_as_uint(tt.y);
    }
  };
  // coefficients (v, u0) and (v, u0+1) for the lane's v: 16 contiguous bytes per (group, v)
  auto store_pair = [&](int u0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const uint32_t g = lane >> 3, v = lane & 7; // 8 groups x 8 coefficient rows = 64 lanes
    const uint32_t c = v * 8 + u0;
    const uint2 lo = *reinterpret_cast<const uint2 *>(wl + c * kQ32RowStride + g * 8);
    const uint2 hi = *reinterpret_cast<const uint2 *>(wl + (c + 1) * kQ32RowStride + g * 8);
    const u32x4_t w = ~u32x4_t{lo.x, lo.y, hi.x, hi.y};
    __builtin_nontemporal_store(w, reinterpret_cast<u32x4_t *>(outw + g * 512 + c * 8));
  };
  using std::integral_constant;
  quant_stage(integral_constant<int, 0>{}); // u = 0, 4
  quant_stage(integral_constant<int, 2>{}); // u = 1, 3
  store_pair(0);
  quant_stage(integral_constant<int, 3>{}); // u = 5, 7
  store_pair(4);
  quant_stage(integral_constant<int, 1>{}); // u = 2, 6
  store_pair(2);
  store_pair(6);
}

// ---------------------------------------------------------------------------------------
// Variant: wave-uniform addressing.  The timeline (profiles/r03_q32_timeline.md) shows the kernel VALU-bound at the
// clock the chip holds under this load (~1.97 GHz): every SIMD retires one wave per ~1.5 us, 2 waves in their compute
// phase saturate it, loads are hidden.  So only fewer VALU cycles per wave help.  Of the product's ~840 VALU
// instructions ~100 are address arithmetic done per lane in 64 bits (v_mad_u64_u32, v_lshl_add_u64, an integer
// division).  Here one workgroup = one wave = one 64-block tile of one block row (2-D grid: x = tile, y = block row,
// needs sizeX % 512 == 0), every base address is wave-uniform (SALU) and the lane contributes a 32-bit offset.
// STAGGER: waves of the first resident generation sleep by their slot number so that each SIMD's first rows arrive
// early instead of all 6 slots' requests queueing behind one another.
// ---------------------------------------------------------------------------------------
template <int MINW, int STAGGER>
__global__ __launch_bounds__(64, MINW) void v_sa64(U8Args a)
{
  __shared__ __attribute__((aligned(16))) uint8_t wl[64 * kQ32RowStride];
  const uint32_t lane = threadIdx.x;
  const uint32_t tile = blockIdx.x, row = blockIdx.y;
  if constexpr (STAGGER > 0)
  {
    if (row * gridDim.x + tile < 6144u)
    {
      const uint32_t slot = GETREG(4) & 15u;
      for (uint32_t i = 0; i < slot; i++)
        __builtin_amdgcn_s_sleep(STAGGER);
    }
  }
  const uint8_t *src = a.from + (size_t)(a.by0 + row) * 8 * a.pitch + (size_t)tile * 512;
  const uint32_t voff = lane * 8;
  uint2 rows[8];
#pragma unroll
  for (int r = 0; r < 8; r++)
    rows[r] = load8_g(sgpr_ptr(src + (size_t)r * a.pitch) + voff);
  uint32_t q[64];
  encode_block_avx_pk<false>(reinterpret_cast<const PkConsts &>(a.pk), rows, a.qt, q);
#pragma unroll
  for (int c = 0; c < 64; c++)
    wl[c * kQ32RowStride + lane] = (uint8_t)q[c];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const gptr_t outw = sgpr_ptr(a.to + ((size_t)(a.by0 + row) * a.bpr + (size_t)tile * 64) * 64);
  const uint32_t rd = (lane & 31) * (2 * kQ32RowStride) + (lane >> 5) * 8; // coefficient rows c2, c2+1 at group 2k + (lane >> 5)
  const uint32_t wr = lane * 16;                                          // == (2k + (lane >> 5)) * 512 + c2 * 8 - k * 1024
#pragma unroll
  for (int k = 0; k < 4; k++)
  {
    const uint2 lo = *reinterpret_cast<const uint2 *>(wl + rd + k * 16);
    const uint2 hi = *reinterpret_cast<const uint2 *>(wl + rd + k * 16 + kQ32RowStride);
    const u32x4_t v = ~u32x4_t{lo.x, lo.y, hi.x, hi.y};
    store16_g(outw + k * 1024 + wr, v);
  }
}

// the same addressing with the product's 4-wave workgroups (1-D grid, a workgroup = 256 consecutive blocks of one block
// row: needs sizeX % 2048 == 0) -- separates the effect of the addressing from that of the dispatch granularity
template <int MINW>
__global__ __launch_bounds__(256, MINW) void v_sa256(U8Args a)
{
  __shared__ __attribute__((aligned(16))) uint8_t lds[4][64 * kQ32RowStride];
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t tile = blockIdx.x * 4 + wave, row = blockIdx.y;
  const uint8_t *src = a.from + (size_t)(a.by0 + row) * 8 * a.pitch + (size_t)tile * 512;
  const uint32_t voff = lane * 8;
  uint2 rows[8];
#pragma unroll
  for (int r = 0; r < 8; r++)
    rows[r] = load8_g(sgpr_ptr(src + (size_t)r * a.pitch) + voff);
  uint32_t q[64];
  encode_block_avx_pk<false>(reinterpret_cast<const PkConsts &>(a.pk), rows, a.qt, q);
  uint8_t *wl = lds[wave];
#pragma unroll
  for (int c = 0; c < 64; c++)
    wl[c * kQ32RowStride + lane] = (uint8_t)q[c];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const gptr_t outw = sgpr_ptr(a.to + ((size_t)(a.by0 + row) * a.bpr + (size_t)tile * 64) * 64);
  const uint32_t rd = (lane & 31) * (2 * kQ32RowStride) + (lane >> 5) * 8;
  const uint32_t wr = lane * 16;
#pragma unroll
  for (int k = 0; k < 4; k++)
  {
    const uint2 lo = *reinterpret_cast<const uint2 *>(wl + rd + k * 16);
    const uint2 hi = *reinterpret_cast<const uint2 *>(wl + rd + k * 16 + kQ32RowStride);
    const u32x4_t v = ~u32x4_t{lo.x, lo.y, hi.x, hi.y};
    store16_g(outw + k * 1024 + wr, v);
  }
}

// ---------------------------------------------------------------------------------------
// What does the LDS reorder cost?  (encq/SSE, 786 VALU and no LDS, runs in 23.7 us; q32, 663 VALU + 64 ds_write_b8, 27.5.)
//   LDSMODE 0  the product tile kernel's reorder: 64 ds_write_b8 + 4 ds_read2_b64 per lane
//   LDSMODE 1  NO reorder (wrong bytes, timing only): the 64 bytes packed with 48 v_perm and stored 4 x 16 B
//   LDSMODE 2  NO reorder, NO packing (wrong bytes): 16 of the 64 words stored as they are -- the transform + quantiser alone
//   LDSMODE 3  byte pairs: coefficients (v,u) and (v,u+4)... merged with one v_perm per pair, 32 ds_write_b16 into rows
//              [pair][lane][2], read back 16 B per lane and de-interleaved with 4 v_perm per 16 B (bit-exact)
// ---------------------------------------------------------------------------------------
template <int LDSMODE>
__global__ __launch_bounds__(64, 6) void v_ldscost(U8Args a)
{
  __shared__ __attribute__((aligned(16))) uint8_t wl[64 * kQ32RowStride * (LDSMODE == 3 ? 2 : 1)];
  const uint32_t lane = threadIdx.x;
  const uint32_t tile = blockIdx.x, row = blockIdx.y;
  uint32_t q[64];
  {
    uint2 rows[8];
    load_block_rows_g(a.from + (size_t)(a.by0 + row) * 8 * a.pitch + (size_t)tile * 512, a.pitch, lane * 8, rows);
    encode_block_avx_pk<false>(reinterpret_cast<const PkConsts &>(a.pk), rows, a.qt, q);
  }
  const gptr_t outw = sgpr_ptr(a.to + ((size_t)(a.by0 + row) * a.bpr + (size_t)tile * 64) * 64);
  if constexpr (LDSMODE == 0)
  {
#pragma unroll
    for (int c = 0; c < 64; c++)
      wl[c * kQ32RowStride + lane] = (uint8_t)q[c];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const uint32_t rd = (lane & 31) * (2 * kQ32RowStride) + (lane >> 5) * 8;
#pragma unroll
    for (int k = 0; k < 4; k++)
    {
      const uint2 lo = *reinterpret_cast<const uint2 *>(wl + rd + k * 16);
      const uint2 hi = *reinterpret_cast<const uint2 *>(wl + rd + k * 16 + kQ32RowStride);
      store16_g(outw + k * 1024 + lane * 16, ~u32x4_g{lo.x, lo.y, hi.x, hi.y});
    }
  }
  else if constexpr (LDSMODE == 1)
  {
#pragma unroll
    for (int k = 0; k < 4; k++)
    {
      u32x4_g v;
#pragma unroll
      for (int j = 0; j < 4; j++)
        v[j] = pack4_lo8(q[k * 16 + j * 4], q[k * 16 + j * 4 + 1], q[k * 16 + j * 4 + 2], q[k * 16 + j * 4 + 3]);
      store16_g(outw + k * 1024 + lane * 16, ~v);
    }
  }
  else if constexpr (LDSMODE == 2)
  {
    uint32_t acc[16];
#pragma unroll
    for (int i = 0; i < 16; i++)
      acc[i] = q[i];
#pragma unroll
    for (int i = 16; i < 64; i++)
      asm volatile("" ::"v"(q[i]));
#pragma unroll
    for (int k = 0; k < 4; k++)
      store16_g(outw + k * 1024 + lane * 16, u32x4_g{acc[4 * k], acc[4 * k + 1], acc[4 * k + 2], acc[4 * k + 3]});
  }
  else
  { // pairs (c, c + 1) with c even: rows [c / 2][lane][2] of 2 * 72 bytes
    constexpr int kRow2 = 2 * kQ32RowStride;
#pragma unroll
    for (int c = 0; c < 64; c += 2)
    {
      const uint32_t w = __builtin_amdgcn_perm(q[c + 1], q[c], 0x0c0c0400u); // byte0 = coef c, byte1 = coef c + 1
      *reinterpret_cast<uint16_t *>(wl + (c / 2) * kRow2 + lane * 2) = (uint16_t)w;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // store k, lane l: coefficients c2 = 2 (l & 31), c2 + 1 of group g = 2k + (l >> 5): pair row l & 31, lanes 8g .. 8g+7 = 16 interleaved bytes
    const uint32_t rd = (lane & 31) * kRow2 + (lane >> 5) * 16;
#pragma unroll
    for (int k = 0; k < 4; k++)
    {
      const uint4 d = *reinterpret_cast<const uint4 *>(wl + rd + k * 32);
      u32x4_g v;
      v[0] = __builtin_amdgcn_perm(d.y, d.x, 0x06040200u); // even bytes: coef c2 of blocks 0..3
      v[1] = __builtin_amdgcn_perm(d.w, d.z, 0x06040200u);
      v[2] = __builtin_amdgcn_perm(d.y, d.x, 0x07050301u); // odd bytes: coef c2 + 1
      v[3] = __builtin_amdgcn_perm(d.w, d.z, 0x07050301u);
      store16_g(outw + k * 1024 + lane * 16, ~v);
    }
  }
}

// the fused int16 round trip (k_i16<MODE_ROUNDTRIP, false>), stamped: 0 entry, 1 all 8 rows arrived, 3 stores issued, 4 acknowledged
// (stamp 2 unused: transform and stores interleave row by row)
__global__ __launch_bounds__(kWG) __attribute__((amdgpu_waves_per_eu(3, 3))) void v_timeline_i16(I16Args a, Stamp *stamps)
{
  STAMP(0);
  const uint32_t t = blockIdx.x * kWG + threadIdx.x;
  const uint32_t row = t / a.bpr;
  const uint32_t bx = t - row * a.bpr;
  const size_t by = a.by0 + row;
  const int16_t *src = a.from + by * 8 * a.pitch_in + (size_t)bx * 8;
  int16_t *dst = a.to + by * 8 * a.pitch_out + (size_t)bx * 8;
  i16_roundtrip_pk<false>(a.consts, src, dst, a.pitch_in, a.pitch_out, 0);
  STAMP(3);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  STAMP(4);
  if ((threadIdx.x & 63) == 0)
  {
    Stamp *s = stamps + ((blockIdx.x * kWG + (threadIdx.x & ~63u)) >> 6);
    s->hw_id = GETREG(4);
    s->xcc_id = GETREG(20);
    s->wave_t0 = blockIdx.x * kWG + (threadIdx.x & ~63u);
    s->pad = 0;
    s->rt[1] = s->rt[2] = 0;
    s->ck[1] = s->ck[2] = 0;
  }
}

// int16 fused round trip as one-wave workgroups on a 2-D grid with wave-uniform addressing, occupancy steered to WAVES
template <int WAVES>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void v_i16_tile(I16Args a)
{
  const size_t by = a.by0 + blockIdx.y;
  const RowsTiled rows{a.from + by * 8 * a.pitch_in + (size_t)blockIdx.x * 512, a.to + by * 8 * a.pitch_out + (size_t)blockIdx.x * 512, a.pitch_in, a.pitch_out, threadIdx.x * 16};
  i16_roundtrip_rows<false>(a.consts, rows, 0);
}
template <int WAVES>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void v_i16_tile_prio(I16Args a)
{
  const size_t by = a.by0 + blockIdx.y;
  const RowsTiled rows{a.from + by * 8 * a.pitch_in + (size_t)blockIdx.x * 512, a.to + by * 8 * a.pitch_out + (size_t)blockIdx.x * 512, a.pitch_in, a.pitch_out, threadIdx.x * 16};
  i16_roundtrip_rows<false, RowsTiled, true>(a.consts, rows, 0);
}
template <int MODE, int WAVES>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void v_i16_tile_mode(I16Args a)
{
  const size_t by = a.by0 + blockIdx.y;
  const RowsTiled rows{a.from + by * 8 * a.pitch_in + (size_t)blockIdx.x * 512, a.to + by * 8 * a.pitch_out + (size_t)blockIdx.x * 512, a.pitch_in, a.pitch_out, threadIdx.x * 16};
  i16_block<MODE, false>(a.consts, rows, a.tb);
}
// the same addressing in 4-wave workgroups (256 consecutive blocks of one block row)
template <int WAVES>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void v_i16_tile256(I16Args a)
{
  const size_t by = a.by0 + blockIdx.y;
  const uint32_t tile = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const RowsTiled rows{a.from + by * 8 * a.pitch_in + (size_t)tile * 512, a.to + by * 8 * a.pitch_out + (size_t)tile * 512, a.pitch_in, a.pitch_out, (threadIdx.x & 63) * 16};
  i16_roundtrip_rows<false>(a.consts, rows, 0);
}

template <int MODE, int WAVES>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void v_f32_tile_prio(F32Args a)
{
  f32_tile_body<MODE, true>(a);
}
template <int MODE, int WAVES>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void v_f32_tile(F32Args a)
{
  f32_tile_body<MODE>(a);
}

// k_q32_tile with phase priorities (1 row pass, 2 column pass + quantiser, 3 read back + stores)
template <int MINW>
__global__ __launch_bounds__(64, MINW) void v_q32_tile_prio(U8Args a)
{
  __shared__ __attribute__((aligned(16))) uint8_t wl[64 * kQ32RowStride];
  const uint32_t lane = threadIdx.x;
  const uint32_t tile = blockIdx.x, row = blockIdx.y;
  uint32_t q[64];
  {
    uint2 rows[8];
    load_block_rows_g(a.from + (size_t)(a.by0 + row) * 8 * a.pitch + (size_t)tile * 512, a.pitch, lane * 8, rows);
    encode_block_avx_pk<false, true>(reinterpret_cast<const PkConsts &>(a.pk), rows, a.qt, q);
  }
#pragma unroll
  for (int c = 0; c < 64; c++)
    wl[c * kQ32RowStride + lane] = (uint8_t)q[c];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  __builtin_amdgcn_s_setprio(3);
  const gptr_t outw = sgpr_ptr(a.to + ((size_t)(a.by0 + row) * a.bpr + (size_t)tile * 64) * 64);
  const uint32_t rd = (lane & 31) * (2 * kQ32RowStride) + (lane >> 5) * 8;
#pragma unroll
  for (int k = 0; k < 4; k++)
  {
    const uint2 lo = *reinterpret_cast<const uint2 *>(wl + rd + k * 16);
    const uint2 hi = *reinterpret_cast<const uint2 *>(wl + rd + k * 16 + kQ32RowStride);
    store16_g(outw + k * 1024 + lane * 16, ~u32x4_g{lo.x, lo.y, hi.x, hi.y});
  }
}

// stereo / SSE tier with a WAVE-private reorder: one workgroup = one wave = 64 consecutive blocks of one (block row, eye) row = 64
// contiguous bytes of each of the 64 coefficient planes (simd_dct.cpp:1061-1099).  Staged as [coef][lane] like q32, then every store
// instruction writes 64-byte pieces of 16 planes.  No workgroup barrier, 4.5 KiB of LDS per wave -- against the product's
// 256-block workgroups with 256-byte pieces behind __syncthreads().
template <int MINW>
__global__ __launch_bounds__(64, MINW) void v_stereo_wave(U8Args a)
{
  __shared__ __attribute__((aligned(16))) uint8_t wl[64 * kQ32RowStride];
  const uint32_t lane = threadIdx.x;
  const uint32_t tile = blockIdx.x, row = blockIdx.y; // row = (block row, eye)
  const uint32_t by = a.by0 + (row >> 1), eye = row & 1;
  uint32_t q[64];
  {
    uint2 rows[8];
    load_block_rows_g(a.from + (size_t)by * 8 * a.pitch + (size_t)eye * a.eye_offset + (size_t)tile * 512, a.pitch, lane * 8, rows);
    encode_block_pk<MDCT_PROFILE_REF_SSE, MDCT_LAYOUT_STEREO, false>(reinterpret_cast<const PkConsts &>(a.pk), rows, a.qt, nullptr, q);
  }
#pragma unroll
  for (int c = 0; c < 64; c++)
    wl[c * kQ32RowStride + lane] = (uint8_t)q[c];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // store k, lane l: plane c = 16 k + (l >> 2), bytes [16 (l & 3), +16) of the wave's 64
  const size_t pos0 = ((size_t)a.by0 * 2 + row) * a.bpr + (size_t)tile * 64;
#pragma unroll
  for (int k = 0; k < 4; k++)
  {
    const uint32_t c = 16 * k + (lane >> 2), part = lane & 3;
    const uint4 v = *reinterpret_cast<const uint4 *>(wl + c * kQ32RowStride + part * 16);
    typedef unsigned int u32x4_unaligned __attribute__((ext_vector_type(4), aligned(1)));
    const u32x4_unaligned w = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(w, reinterpret_cast<u32x4_unaligned *>(a.to + a.plane_stride * c + pos0 + part * 16));
  }
}

// ---------------------------------------------------------------------------------------
// k_q32_tile over NT consecutive tiles of a block row per wave, the next tile's eight row loads issued between the row pass and
// the column pass of the current one (the 16 registers of the raw rows are dead by then: no extra register pressure): every
// wave after its first tile finds its rows waiting -- in-wave prefetch on top of the 6 waves per SIMD.
// ---------------------------------------------------------------------------------------
template <int NT, int MINW>
__global__ __launch_bounds__(64, MINW) void v_tiles(U8Args a)
{
  __shared__ __attribute__((aligned(16))) uint8_t wl[64 * kQ32RowStride];
  const uint32_t lane = threadIdx.x;
  const uint32_t tile0 = blockIdx.x * NT, row = blockIdx.y;
  const uint8_t *src = a.from + (size_t)(a.by0 + row) * 8 * a.pitch + (size_t)tile0 * 512;
  uint2 rows[8], nxt[8];
  load_block_rows_g(src, a.pitch, lane * 8, rows);
#pragma unroll
  for (int t = 0; t < NT; t++)
  {
    uint32_t q[64];
    if (t + 1 < NT)
      encode_block_avx_pk<false>(reinterpret_cast<const PkConsts &>(a.pk), rows, a.qt, q, [&]() { load_block_rows_g(src + (size_t)(t + 1) * 512, a.pitch, lane * 8, nxt); });
    else
      encode_block_avx_pk<false>(reinterpret_cast<const PkConsts &>(a.pk), rows, a.qt, q);
#pragma unroll
    for (int c = 0; c < 64; c++)
      wl[c * kQ32RowStride + lane] = (uint8_t)q[c];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const gptr_t outw = sgpr_ptr(a.to + ((size_t)(a.by0 + row) * a.bpr + (size_t)(tile0 + t) * 64) * 64);
    const uint32_t rd = (lane & 31) * (2 * kQ32RowStride) + (lane >> 5) * 8;
#pragma unroll
    for (int k = 0; k < 4; k++)
    {
      const uint2 lo = *reinterpret_cast<const uint2 *>(wl + rd + k * 16);
      const uint2 hi = *reinterpret_cast<const uint2 *>(wl + rd + k * 16 + kQ32RowStride);
      store16_g(outw + k * 1024 + lane * 16, ~u32x4_g{lo.x, lo.y, hi.x, hi.y});
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier(); // the staging buffer is read before the next tile overwrites it
#pragma unroll
    for (int r = 0; r < 8; r++)
      rows[r] = nxt[r];
  }
}

// ---------------------------------------------------------------------------------------
// The review's variant (i): a specialised LOADER wave per workgroup.  Workgroup = 1 loader wave + NC compute waves, one block
// row (sizeX / 512 tiles) per workgroup in generations of NC tiles.  The loader issues global_load_lds_dwordx4 (64 lanes x 16 B
// = two 512-byte pixel rows of a tile per instruction, straight into an LDS ring of two generations, no VGPRs), waits for the
// generation the compute waves are about to read, and meets them at a workgroup barrier; a compute wave reads its block's rows by
// ds_read_b64 (conflict-free), runs the product's encode_block_avx_pk and stages / stores its tile like k_q32_tile.
// Two ring slots: generation g + 1 is issued right after barrier g (every compute wave is then through with generation g - 1,
// the slot's previous tenant) and has the whole of generation g's arithmetic to land.
// ---------------------------------------------------------------------------------------
template <int NC, int MINW, int SPLIT>
__global__ __launch_bounds__(64 * (NC + 1), MINW) void v_loader(U8Args a)
{
  // a slot holds a tile's 8 x 512 input bytes and, once its compute wave has them in registers, the same tile's output
  // staged as [coef][lane] (64 x kQ32RowStride = 4608 bytes): 9 KiB of LDS per compute wave
  __shared__ __attribute__((aligned(16))) uint8_t ring[2][NC][64 * kQ32RowStride];
  const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint32_t row = blockIdx.x / SPLIT, part = blockIdx.x % SPLIT;
  const uint32_t tiles = a.bpr / 64 / SPLIT, gens = tiles / NC, tile0 = part * tiles;
  const uint8_t *src_row = a.from + (size_t)(a.by0 + row) * 8 * a.pitch + (size_t)tile0 * 512;
  if (w == NC)
  { // the loader
    const uint32_t half = lane >> 5, l32 = lane & 31;
    auto issue = [&](uint32_t g) {
#pragma unroll
      for (int t = 0; t < NC; t++)
      {
        const uint8_t *src16 = src_row + (size_t)half * a.pitch + (size_t)(g * NC + t) * 512 + l32 * 16;
#pragma unroll
        for (int k = 0; k < 4; k++)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src16 + (size_t)(2 * k) * a.pitch),
                                           (__attribute__((address_space(3))) void *)(&ring[g & 1][t][k * 1024]), 16, 0, 0);
      }
    };
    issue(0);
    for (uint32_t g = 0; g < gens; g++)
    {
      __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0): generation g has landed
      __builtin_amdgcn_s_barrier();       // ... and every compute wave is through with generation g - 1, whose slot is free now
      if (g + 1 < gens)
        issue(g + 1); // in flight while the compute waves work on generation g
    }
    return;
  }
  for (uint32_t g = 0; g < gens; g++)
  {
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    // read by inline asm: the compiler would put s_waitcnt vmcnt(0) -- i.e. a wait for this wave's own stores of the previous
    // generation -- before every LDS read of a buffer that an LDS-DMA instruction anywhere in the kernel may write
    uint8_t *wl = ring[g & 1][w];
    const uint32_t slot_addr = (uint32_t)(uintptr_t)wl;
    const uint32_t in_addr = slot_addr + lane * 8;
    uint2 rows[8];
    unsigned long long rr[8];
#pragma unroll
    for (int r = 0; r < 8; r++)
      asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(rr[r]) : "v"(in_addr), "n"(r * 512) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rr[0]), "+v"(rr[1]), "+v"(rr[2]), "+v"(rr[3]), "+v"(rr[4]), "+v"(rr[5]), "+v"(rr[6]), "+v"(rr[7])::"memory");
#pragma unroll
    for (int r = 0; r < 8; r++)
      rows[r] = uint2{(uint32_t)rr[r], (uint32_t)(rr[r] >> 32)};
    uint32_t q[64];
    encode_block_avx_pk<false>(reinterpret_cast<const PkConsts &>(a.pk), rows, a.qt, q);
    // the output goes through the slot the input came from (asm again, for the same reason)
    const uint32_t wr_addr = slot_addr + lane;
#pragma unroll
    for (int c = 0; c < 64; c++)
      asm volatile("ds_write_b8 %0, %1 offset:%2" ::"v"(wr_addr), "v"(q[c]), "n"(c * kQ32RowStride) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    const uint32_t tile = tile0 + g * NC + w;
    const gptr_t outw = sgpr_ptr(a.to + ((size_t)(a.by0 + row) * a.bpr + (size_t)tile * 64) * 64);
    const uint32_t rd_addr = slot_addr + (lane & 31) * (2 * kQ32RowStride) + (lane >> 5) * 8;
    unsigned long long lo[4], hi[4];
#pragma unroll
    for (int k = 0; k < 4; k++)
    {
      asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(lo[k]) : "v"(rd_addr), "n"(k * 16) : "memory");
      asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(hi[k]) : "v"(rd_addr), "n"(k * 16 + kQ32RowStride) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(lo[3]), "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]), "+v"(hi[3])::"memory");
#pragma unroll
    for (int k = 0; k < 4; k++)
      store16_g(outw + k * 1024 + lane * 16, ~u32x4_g{(uint32_t)lo[k], (uint32_t)(lo[k] >> 32), (uint32_t)hi[k], (uint32_t)(hi[k] >> 32)});
  }
}

int main(int argc, char **argv)
{
  const char *mode = argc > 1 ? argv[1] : "ab";
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
  float q[64];
  for (int i = 0; i < 64; i++) q[i] = 255.0f / ((0.1f + 0.01f * i) * 2000 * 0.95f);
  a.pitch = W; a.sizeX = W; a.out_strip = 8 * W; a.out_tight = 1; a.bpr = W / 8; a.by0 = 0; a.nblocks = (uint32_t)(W / 8 * H / 8);
  const float magicC = 12582912.0f + 128.0f;
  for (int v = 0; v < 8; v++)
    for (int j = 0; j < 4; j++)
    {
      a.qt.q[(v * 4 + j) * 2] = -q[v * 8 + kPairA[j]];
      a.qt.q[(v * 4 + j) * 2 + 1] = -q[v * 8 + kPairB[j]];
    }
  a.pk = PkConstsArg{{a.consts.a, a.consts.f}, {a.consts.c, a.consts.d}, {a.consts.b, a.consts.e}, {a.consts.n, magicC}};
  auto args = [&](int s) { U8Args x = a; x.from = A[s]; x.to = B[s]; return x; };
  const uint32_t nwg = a.nblocks / 256, nwaves = a.nblocks / 64;

  if (!strcmp(mode, "timeline"))
  {
    Stamp *d;
    hipMalloc(&d, sizeof(Stamp) * nwaves);
    // steady state first (power management), then ONE stamped launch directly behind product launches
    for (int i = 0; i < 1500; i++) launch_fwd_quant_u8(args(i % NS), MDCT_LAYOUT_Q32, MDCT_PROFILE_REF_AVX, false, 0);
    for (int rep = 0; rep < 3; rep++)
    {
      for (int i = 0; i < 50; i++) launch_fwd_quant_u8(args(i % NS), MDCT_LAYOUT_Q32, MDCT_PROFILE_REF_AVX, false, 0);
      hipLaunchKernelGGL((v_timeline<6>), dim3(nwg), dim3(256), 0, 0, args(rep % NS), d);
    }
    hipDeviceSynchronize();
    std::vector<Stamp> hs(nwaves);
    hipMemcpy(hs.data(), d, sizeof(Stamp) * nwaves, hipMemcpyDeviceToHost);
    const char *path = argc > 2 ? argv[2] : "gpurun_out/q32_timeline.bin";
    FILE *f = fopen(path, "wb");
    if (!f) { perror(path); return 1; }
    fwrite(hs.data(), sizeof(Stamp), nwaves, f);
    fclose(f);
    // bytes of the stamped kernel == product
    std::vector<uint8_t> ref(bytes), got(bytes);
    launch_fwd_quant_u8(args(0), MDCT_LAYOUT_Q32, MDCT_PROFILE_REF_AVX, false, 0);
    hipMemcpy(ref.data(), B[0], bytes, hipMemcpyDeviceToHost);
    hipMemset(B[0], 0x55, bytes);
    hipLaunchKernelGGL((v_timeline<6>), dim3(nwg), dim3(256), 0, 0, args(0), d);
    hipMemcpy(got.data(), B[0], bytes, hipMemcpyDeviceToHost);
    printf("timeline kernel %s; %u waves -> %s\n", memcmp(ref.data(), got.data(), bytes) ? "!! MISMATCH" : "bit-exact", nwaves, path);
    return 0;
  }

  if (!strcmp(mode, "ab_stereo"))
  {
    U8Args st = a;
    for (int i = 0; i < 64; i++) q[i] = 255.0f / ((0.1f + 0.01f * i) * 8 * 0.95f);
    for (int m = 0; m < 8; m++)
      for (int j = 0; j < 4; j++)
      { // stereo: columns first, pair (m, j) = coefficients (kPairA[j], m) / (kPairB[j], m) stored at first*8 + second
        st.qt.q[(m * 4 + j) * 2] = q[kPairA[j] * 8 + m];
        st.qt.q[(m * 4 + j) * 2 + 1] = q[kPairB[j] * 8 + m];
      }
    st.pk = PkConstsArg{{a.consts.a, a.consts.f}, {a.consts.c, a.consts.d}, {a.consts.b, a.consts.e}, {a.consts.n, a.consts.magic23},
                        {a.consts.d, a.consts.a}, {a.consts.f, a.consts.d}, {a.consts.f, a.consts.c}, {a.consts.c, a.consts.a}, {1.f / (float)0xFF, 127.0f}};
    st.eye_offset = W * (H / 2);
    st.plane_stride = W * H / 64;
    st.nblocks = (uint32_t)(W / 8 * H / 8);
    auto sargs = [&](int s) { U8Args x = st; x.from = A[s]; x.to = B[s]; return x; };
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    struct V { const char *name; std::function<void(int)> f; std::vector<float> t; };
    std::vector<V> vs;
    const dim3 g64((unsigned)(W / 512), (unsigned)(H / 8));
    vs.push_back({"stereo/SSE product (256-block workgroups)", [&](int s) { launch_fwd_quant_u8(sargs(s), MDCT_LAYOUT_STEREO, MDCT_PROFILE_REF_SSE, false, 0); }, {}});
    vs.push_back({"stereo/SSE wave-private reorder 6w", [&](int s) { hipLaunchKernelGGL((v_stereo_wave<6>), g64, dim3(64), 0, 0, sargs(s)); }, {}});
    vs.push_back({"stereo/SSE wave-private reorder 5w", [&](int s) { hipLaunchKernelGGL((v_stereo_wave<5>), g64, dim3(64), 0, 0, sargs(s)); }, {}});
    std::vector<uint8_t> ref(bytes), got(bytes);
    vs[0].f(0); hipMemcpy(ref.data(), B[0], bytes, hipMemcpyDeviceToHost);
    for (size_t i = 1; i < vs.size(); i++)
    {
      hipMemset(B[0], 0x55, bytes);
      vs[i].f(0);
      hipMemcpy(got.data(), B[0], bytes, hipMemcpyDeviceToHost);
      printf("%-44s %s\n", vs[i].name, memcmp(ref.data(), got.data(), bytes) ? "!! MISMATCH" : "bit-exact");
    }
    for (auto &v : vs) for (int i = 0; i < 300; i++) v.f(i % NS);
    hipDeviceSynchronize();
    for (int round = 0; round < 11; round++)
      for (auto &v : vs)
      {
        for (int i = 0; i < 40; i++) v.f(i % NS);
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
      printf("%-44s median %7.2f us  min %7.2f us\n", v.name, v.t[v.t.size() / 2] * 1e3, v.t[0] * 1e3);
    }
    return 0;
  }
  if (!strcmp(mode, "ab_f32"))
  {
    const size_t b32 = W * H * 4;
    float *S[2], *D[2];
    std::vector<float> hf(W * H);
    for (size_t i = 0; i < W * H; i++) hf[i] = (float)host[i] - 128.0f;
    for (int i = 0; i < 2; i++)
    {
      if (hipMalloc(&S[i], b32) != hipSuccess || hipMalloc(&D[i], b32) != hipSuccess) { puts("alloc failed"); return 1; }
      hipMemcpy(S[i], hf.data(), b32, hipMemcpyHostToDevice);
    }
    F32Args fa;
    memset(&fa, 0, sizeof(fa));
    fa.consts = DctConsts();
    for (int i = 0; i < 64; i++) fa.scale[i] = 0.125f + 0.001f * i;
    fa.pitch_in = fa.pitch_out = W; fa.bpr = W / 8; fa.by0 = 0; fa.nblocks = (uint32_t)(W / 8 * H / 8);
    auto fargs = [&](int s) { F32Args x = fa; x.from = S[s & 1]; x.to = D[s & 1]; return x; };
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    struct V { const char *name; std::function<void(int)> f; std::vector<float> t; };
    std::vector<V> vs;
    const dim3 g64((unsigned)(W / 512), (unsigned)(H / 8));
    vs.push_back({"k_f32<FWD, WIDE> linear 256", [&](int s) { hipLaunchKernelGGL((k_f32<MODE_FWD, true>), dim3(nwg), dim3(256), 0, 0, fargs(s)); }, {}});
    vs.push_back({"f32 fwd tile, 2 waves", [&](int s) { hipLaunchKernelGGL((v_f32_tile<MODE_FWD, 2>), g64, dim3(64), 0, 0, fargs(s)); }, {}});
    vs.push_back({"f32 fwd tile, 2 waves, phase prio", [&](int s) { hipLaunchKernelGGL((v_f32_tile_prio<MODE_FWD, 2>), g64, dim3(64), 0, 0, fargs(s)); }, {}});
    vs.push_back({"f32 fwd tile, 3 waves, phase prio", [&](int s) { hipLaunchKernelGGL((v_f32_tile_prio<MODE_FWD, 3>), g64, dim3(64), 0, 0, fargs(s)); }, {}});
    vs.push_back({"f32 inv tile, 2 waves, phase prio", [&](int s) { hipLaunchKernelGGL((v_f32_tile_prio<MODE_INV, 2>), g64, dim3(64), 0, 0, fargs(s)); }, {}});
    vs.push_back({"f32 fwd tile, 3 waves", [&](int s) { hipLaunchKernelGGL((v_f32_tile<MODE_FWD, 3>), g64, dim3(64), 0, 0, fargs(s)); }, {}});
    vs.push_back({"f32 fwd tile, 4 waves", [&](int s) { hipLaunchKernelGGL((v_f32_tile<MODE_FWD, 4>), g64, dim3(64), 0, 0, fargs(s)); }, {}});
    vs.push_back({"k_f32<INV, WIDE> linear 256", [&](int s) { hipLaunchKernelGGL((k_f32<MODE_INV, true>), dim3(nwg), dim3(256), 0, 0, fargs(s)); }, {}});
    vs.push_back({"f32 inv tile, 2 waves", [&](int s) { hipLaunchKernelGGL((v_f32_tile<MODE_INV, 2>), g64, dim3(64), 0, 0, fargs(s)); }, {}});
    vs.push_back({"f32 inv tile, 3 waves", [&](int s) { hipLaunchKernelGGL((v_f32_tile<MODE_INV, 3>), g64, dim3(64), 0, 0, fargs(s)); }, {}});
    vs.push_back({"f32 inv tile, 4 waves", [&](int s) { hipLaunchKernelGGL((v_f32_tile<MODE_INV, 4>), g64, dim3(64), 0, 0, fargs(s)); }, {}});
    vs.push_back({"stream copy of the same bytes", [&](int s) { launch_stream_copy(S[s & 1], D[s & 1], b32, 256, 0); }, {}});
    std::vector<uint8_t> ref(b32), got(b32);
    for (int base : {0, 4})
    {
      vs[base].f(0); hipMemcpy(ref.data(), D[0], b32, hipMemcpyDeviceToHost);
      for (int i = base + 1; i < base + 4; i++)
      {
        hipMemset(D[0], 0x55, b32);
        vs[i].f(0);
        hipMemcpy(got.data(), D[0], b32, hipMemcpyDeviceToHost);
        printf("%-34s %s\n", vs[i].name, memcmp(ref.data(), got.data(), b32) ? "!! MISMATCH" : "bit-exact");
      }
    }
    for (auto &v : vs) for (int i = 0; i < 200; i++) v.f(i);
    hipDeviceSynchronize();
    for (int round = 0; round < 9; round++)
      for (auto &v : vs)
      {
        for (int i = 0; i < 30; i++) v.f(i);
        hipEventRecord(e0, 0);
        for (int i = 0; i < 30; i++) v.f(i);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        v.t.push_back(ms / 30);
      }
    for (auto &v : vs)
    {
      std::sort(v.t.begin(), v.t.end());
      printf("%-34s median %7.2f us  min %7.2f us\n", v.name, v.t[v.t.size() / 2] * 1e3, v.t[0] * 1e3);
    }
    return 0;
  }
  if (!strcmp(mode, "timeline_i16") || !strcmp(mode, "ab_i16"))
  {
    const size_t b16 = W * H * 2;
    int16_t *S[NS], *D[NS];
    for (int i = 0; i < NS; i++)
    {
      if (hipMalloc(&S[i], b16) != hipSuccess || hipMalloc(&D[i], b16) != hipSuccess) { puts("alloc failed"); return 1; }
      hipMemcpy(S[i], host.data(), bytes, hipMemcpyHostToDevice);
      hipMemcpy(reinterpret_cast<uint8_t *>(S[i]) + bytes, host.data(), bytes, hipMemcpyHostToDevice);
    }
    I16Args ia;
    memset(&ia, 0, sizeof(ia));
    ia.consts = DctConsts();
    ia.pitch_in = ia.pitch_out = W; ia.bpr = W / 8; ia.by0 = 0; ia.nblocks = (uint32_t)(W / 8 * H / 8);
    auto iargs = [&](int s) { I16Args x = ia; x.from = S[s]; x.to = D[s]; return x; };
    if (!strcmp(mode, "ab_i16"))
    {
      hipEvent_t e0, e1;
      hipEventCreate(&e0); hipEventCreate(&e1);
      struct V { const char *name; std::function<void(int)> f; std::vector<float> t; };
      std::vector<V> vs;
      const dim3 g64((unsigned)(W / 512), (unsigned)(H / 8)), g256((unsigned)(W / 2048), (unsigned)(H / 8));
      vs.push_back({"launch_i16 (library default)", [&](int s) { launch_i16(iargs(s), MODE_ROUNDTRIP, false, 0); }, {}});
      vs.push_back({"k_i16<ROUNDTRIP> linear 256", [&](int s) { hipLaunchKernelGGL((k_i16<MODE_ROUNDTRIP, false>), dim3(nwg), dim3(256), 0, 0, iargs(s)); }, {}});
      vs.push_back({"tile 64 thr, 2 waves", [&](int s) { hipLaunchKernelGGL((v_i16_tile<2>), g64, dim3(64), 0, 0, iargs(s)); }, {}});
      vs.push_back({"tile 64 thr, 3 waves", [&](int s) { hipLaunchKernelGGL((v_i16_tile<3>), g64, dim3(64), 0, 0, iargs(s)); }, {}});
      vs.push_back({"tile 64 thr, 4 waves", [&](int s) { hipLaunchKernelGGL((v_i16_tile<4>), g64, dim3(64), 0, 0, iargs(s)); }, {}});
      vs.push_back({"tile 64 thr, 5 waves", [&](int s) { hipLaunchKernelGGL((v_i16_tile<5>), g64, dim3(64), 0, 0, iargs(s)); }, {}});
      vs.push_back({"tile 64 thr, 6 waves", [&](int s) { hipLaunchKernelGGL((v_i16_tile<6>), g64, dim3(64), 0, 0, iargs(s)); }, {}});
      vs.push_back({"tile 64 thr, 2 waves, phase prio", [&](int s) { hipLaunchKernelGGL((v_i16_tile_prio<2>), g64, dim3(64), 0, 0, iargs(s)); }, {}});
      vs.push_back({"tile 64 thr, 3 waves, phase prio", [&](int s) { hipLaunchKernelGGL((v_i16_tile_prio<3>), g64, dim3(64), 0, 0, iargs(s)); }, {}});
      vs.push_back({"tile 64 thr, 4 waves, phase prio", [&](int s) { hipLaunchKernelGGL((v_i16_tile_prio<4>), g64, dim3(64), 0, 0, iargs(s)); }, {}});
      vs.push_back({"tile 64 thr, 6 waves, phase prio", [&](int s) { hipLaunchKernelGGL((v_i16_tile_prio<6>), g64, dim3(64), 0, 0, iargs(s)); }, {}});
      vs.push_back({"tile 256 thr, 3 waves", [&](int s) { hipLaunchKernelGGL((v_i16_tile256<3>), g256, dim3(256), 0, 0, iargs(s)); }, {}});
      vs.push_back({"tile 256 thr, 4 waves", [&](int s) { hipLaunchKernelGGL((v_i16_tile256<4>), g256, dim3(256), 0, 0, iargs(s)); }, {}});
      const size_t n_rt = vs.size();
      vs.push_back({"fwd: k_i16<FWD> linear 256", [&](int s) { hipLaunchKernelGGL((k_i16<MODE_FWD, false>), dim3(nwg), dim3(256), 0, 0, iargs(s)); }, {}});
      vs.push_back({"fwd: tile 64 thr, 2 waves", [&](int s) { hipLaunchKernelGGL((v_i16_tile_mode<MODE_FWD, 2>), g64, dim3(64), 0, 0, iargs(s)); }, {}});
      vs.push_back({"fwd: tile 64 thr, 3 waves", [&](int s) { hipLaunchKernelGGL((v_i16_tile_mode<MODE_FWD, 3>), g64, dim3(64), 0, 0, iargs(s)); }, {}});
      vs.push_back({"fwd: tile 64 thr, 4 waves", [&](int s) { hipLaunchKernelGGL((v_i16_tile_mode<MODE_FWD, 4>), g64, dim3(64), 0, 0, iargs(s)); }, {}});
      vs.push_back({"fwd: tile 64 thr, 6 waves", [&](int s) { hipLaunchKernelGGL((v_i16_tile_mode<MODE_FWD, 6>), g64, dim3(64), 0, 0, iargs(s)); }, {}});
      vs.push_back({"inv: k_i16<INV> linear 256", [&](int s) { hipLaunchKernelGGL((k_i16<MODE_INV, false>), dim3(nwg), dim3(256), 0, 0, iargs(s)); }, {}});
      vs.push_back({"inv: tile 64 thr, 2 waves", [&](int s) { hipLaunchKernelGGL((v_i16_tile_mode<MODE_INV, 2>), g64, dim3(64), 0, 0, iargs(s)); }, {}});
      vs.push_back({"inv: tile 64 thr, 3 waves", [&](int s) { hipLaunchKernelGGL((v_i16_tile_mode<MODE_INV, 3>), g64, dim3(64), 0, 0, iargs(s)); }, {}});
      vs.push_back({"inv: tile 64 thr, 4 waves", [&](int s) { hipLaunchKernelGGL((v_i16_tile_mode<MODE_INV, 4>), g64, dim3(64), 0, 0, iargs(s)); }, {}});
      vs.push_back({"inv: tile 64 thr, 6 waves", [&](int s) { hipLaunchKernelGGL((v_i16_tile_mode<MODE_INV, 6>), g64, dim3(64), 0, 0, iargs(s)); }, {}});
      vs.push_back({"stream copy (roofline)", [&](int s) { launch_stream_copy(S[s], D[s], b16, 256, 0); }, {}});
      std::vector<uint8_t> ref(b16), got(b16);
      vs[1].f(0); hipMemcpy(ref.data(), D[0], b16, hipMemcpyDeviceToHost);
      for (size_t i = 0; i < n_rt; i++)
      {
        hipMemset(D[0], 0x55, b16);
        vs[i].f(0);
        hipMemcpy(got.data(), D[0], b16, hipMemcpyDeviceToHost);
        printf("%-30s %s\n", vs[i].name, memcmp(ref.data(), got.data(), b16) ? "!! MISMATCH" : "bit-exact");
      }
      for (auto &v : vs) for (int i = 0; i < 300; i++) v.f(i % NS);
      hipDeviceSynchronize();
      for (int round = 0; round < 15; round++)
        for (auto &v : vs)
        {
          for (int i = 0; i < 40; i++) v.f(i % NS);
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
        printf("%-34s median %7.2f us  min %7.2f us  max %7.2f us\n", v.name, v.t[v.t.size() / 2] * 1e3, v.t[0] * 1e3, v.t.back() * 1e3);
      }
      return 0;
    }
    Stamp *d;
    hipMalloc(&d, sizeof(Stamp) * nwaves);
    hipMemset(d, 0, sizeof(Stamp) * nwaves);
    for (int i = 0; i < 1200; i++) launch_i16(iargs(i % NS), MODE_ROUNDTRIP, false, 0);
    for (int rep = 0; rep < 3; rep++)
    {
      for (int i = 0; i < 50; i++) launch_i16(iargs(i % NS), MODE_ROUNDTRIP, false, 0);
      hipLaunchKernelGGL(v_timeline_i16, dim3(nwg), dim3(256), 0, 0, iargs(rep % NS), d);
    }
    hipDeviceSynchronize();
    std::vector<Stamp> hs(nwaves);
    hipMemcpy(hs.data(), d, sizeof(Stamp) * nwaves, hipMemcpyDeviceToHost);
    const char *path = argc > 2 ? argv[2] : "gpurun_out/i16_timeline.bin";
    FILE *f = fopen(path, "wb");
    if (!f) { perror(path); return 1; }
    fwrite(hs.data(), sizeof(Stamp), nwaves, f);
    fclose(f);
    printf("i16 round-trip timeline: %u waves -> %s\n", nwaves, path);
    return 0;
  }

  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  struct V { const char *name; std::function<void(int)> f; std::vector<float> t; bool check; };
  std::vector<V> vs;
  vs.push_back({"product q32", [&](int s) { launch_fwd_quant_u8(args(s), MDCT_LAYOUT_Q32, MDCT_PROFILE_REF_AVX, false, 0); }, {}, false});
  const dim3 g64((unsigned)(a.bpr / 64), (unsigned)(H / 8)), g256((unsigned)(a.bpr / 256), (unsigned)(H / 8));
  vs.push_back({"sa64 6w", [&](int s) { hipLaunchKernelGGL((v_sa64<6, 0>), g64, dim3(64), 0, 0, args(s)); }, {}, true});
  vs.push_back({"sa64 5w", [&](int s) { hipLaunchKernelGGL((v_sa64<5, 0>), g64, dim3(64), 0, 0, args(s)); }, {}, true});
  vs.push_back({"sa64 4w", [&](int s) { hipLaunchKernelGGL((v_sa64<4, 0>), g64, dim3(64), 0, 0, args(s)); }, {}, true});
  vs.push_back({"sa64 6w stagger 4", [&](int s) { hipLaunchKernelGGL((v_sa64<6, 4>), g64, dim3(64), 0, 0, args(s)); }, {}, true});
  vs.push_back({"sa64 6w stagger 8", [&](int s) { hipLaunchKernelGGL((v_sa64<6, 8>), g64, dim3(64), 0, 0, args(s)); }, {}, true});
  vs.push_back({"sa256 6w", [&](int s) { hipLaunchKernelGGL((v_sa256<6>), g256, dim3(256), 0, 0, args(s)); }, {}, true});
  vs.push_back({"sa256 5w", [&](int s) { hipLaunchKernelGGL((v_sa256<5>), g256, dim3(256), 0, 0, args(s)); }, {}, true});
  vs.push_back({"q32 tile (product kernel)", [&](int s) { hipLaunchKernelGGL(k_q32_tile, g64, dim3(64), 0, 0, args(s)); }, {}, true});
  vs.push_back({"q32 tile, phase prio 6w", [&](int s) { hipLaunchKernelGGL((v_q32_tile_prio<6>), g64, dim3(64), 0, 0, args(s)); }, {}, true});
  vs.push_back({"q32 tile, phase prio 5w", [&](int s) { hipLaunchKernelGGL((v_q32_tile_prio<5>), g64, dim3(64), 0, 0, args(s)); }, {}, true});
  vs.push_back({"2 tiles per wave, prefetch, 6w", [&](int s) { hipLaunchKernelGGL((v_tiles<2, 6>), dim3((unsigned)(a.bpr / 128), (unsigned)(H / 8)), dim3(64), 0, 0, args(s)); }, {}, true});
  vs.push_back({"2 tiles per wave, prefetch, 5w", [&](int s) { hipLaunchKernelGGL((v_tiles<2, 5>), dim3((unsigned)(a.bpr / 128), (unsigned)(H / 8)), dim3(64), 0, 0, args(s)); }, {}, true});
  vs.push_back({"4 tiles per wave, prefetch, 6w", [&](int s) { hipLaunchKernelGGL((v_tiles<4, 6>), dim3((unsigned)(a.bpr / 256), (unsigned)(H / 8)), dim3(64), 0, 0, args(s)); }, {}, true});
  vs.push_back({"4 tiles per wave, prefetch, 5w", [&](int s) { hipLaunchKernelGGL((v_tiles<4, 5>), dim3((unsigned)(a.bpr / 256), (unsigned)(H / 8)), dim3(64), 0, 0, args(s)); }, {}, true});
  vs.push_back({"16 tiles per wave (a row), 6w", [&](int s) { hipLaunchKernelGGL((v_tiles<16, 6>), dim3((unsigned)(a.bpr / 1024), (unsigned)(H / 8)), dim3(64), 0, 0, args(s)); }, {}, true});
  vs.push_back({"loader + 4 compute, 1 WG/row, 5w", [&](int s) { hipLaunchKernelGGL((v_loader<4, 5, 1>), dim3((unsigned)(H / 8)), dim3(320), 0, 0, args(s)); }, {}, true});
  vs.push_back({"loader + 4 compute, 2 WG/row, 5w", [&](int s) { hipLaunchKernelGGL((v_loader<4, 5, 2>), dim3((unsigned)(H / 8) * 2), dim3(320), 0, 0, args(s)); }, {}, true});
  vs.push_back({"loader + 2 compute, 2 WG/row, 6w", [&](int s) { hipLaunchKernelGGL((v_loader<2, 6, 2>), dim3((unsigned)(H / 8) * 2), dim3(192), 0, 0, args(s)); }, {}, true});
  vs.push_back({"loader + 2 compute, 4 WG/row, 6w", [&](int s) { hipLaunchKernelGGL((v_loader<2, 6, 4>), dim3((unsigned)(H / 8) * 4), dim3(192), 0, 0, args(s)); }, {}, true});
  vs.push_back({"loader + 1 compute, 4 WG/row, 6w", [&](int s) { hipLaunchKernelGGL((v_loader<1, 6, 4>), dim3((unsigned)(H / 8) * 4), dim3(128), 0, 0, args(s)); }, {}, true});
  vs.push_back({"tile: lds b8 (product)", [&](int s) { hipLaunchKernelGGL((v_ldscost<0>), g64, dim3(64), 0, 0, args(s)); }, {}, true});
  vs.push_back({"tile: no lds, 48 perm", [&](int s) { hipLaunchKernelGGL((v_ldscost<1>), g64, dim3(64), 0, 0, args(s)); }, {}, false});
  vs.push_back({"tile: no lds, no pack", [&](int s) { hipLaunchKernelGGL((v_ldscost<2>), g64, dim3(64), 0, 0, args(s)); }, {}, false});
  vs.push_back({"tile: lds b16 pairs", [&](int s) { hipLaunchKernelGGL((v_ldscost<3>), g64, dim3(64), 0, 0, args(s)); }, {}, true});
  vs.push_back({"early stores 6w", [&](int s) { hipLaunchKernelGGL((v_early<6>), dim3(nwg), dim3(256), 0, 0, args(s)); }, {}, true});
  vs.push_back({"early stores 5w", [&](int s) { hipLaunchKernelGGL((v_early<5>), dim3(nwg), dim3(256), 0, 0, args(s)); }, {}, true});
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
  for (auto &v : vs) for (int i = 0; i < 300; i++) v.f(i % NS);
  hipDeviceSynchronize();
  for (int round = 0; round < 7; round++)
    for (auto &v : vs)
    {
      for (int i = 0; i < 40; i++) v.f(i % NS);
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
