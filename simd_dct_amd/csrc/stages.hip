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

#include "wg_sync.h"
#include "huffman_rows.h"
#include "pack_rows.h"
#include "mdct.h"
#include "scan_records.h"

extern "C" __attribute__((visibility("hidden"))) int mdct_set_error(int code, const char *fmt, ...); // mdct_api.hip

namespace mdct
{

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
// k_scan: every wave works alone (wave-private LDS records), so the workgroup size only sets the LDS granule and the
// dispatch rate; measured per source (profiles/r02_scan_workgroup_size.log)
constexpr int scan_wg(int src) { return src == 2 ? 128 : 64; } // SRC_STEREO : SRC_I16, SRC_Q32, SRC_BLOCK

template <int SRC, bool RLE>
__global__ __launch_bounds__(scan_wg(SRC)) void k_scan(ScanArgs a)
{
  constexpr int kScanWG = scan_wg(SRC);
  __shared__ __attribute__((aligned(16))) uint8_t lv_all[kScanWG / 64][64 * kLvRow];
  __shared__ __attribute__((aligned(16))) uint8_t rn_all[kScanWG / 64][RLE ? 64 * kRnRow : 16];
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t wave_t0 = blockIdx.x * kScanWG + wave * 64; // first block of the wave within the launch
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
  { // the wave's 64 blocks are 8 consecutive q32 groups [group][coef][8 blocks] = 4 KiB: coefficient c of the lane's block is
    // one byte load, 8 x 8 contiguous bytes per wave instruction, all 64 instructions inside the same 32 cache lines.
    // (Staging the 4 KiB through LDS with 16 B per lane loads and reading it back bytewise was 20 us slower.)
    const uint8_t *p = static_cast<const uint8_t *>(a.src) + blk0 * 64 + (size_t)((valid ? lane : 0) >> 3) * 512 + ((valid ? lane : 0) & 7);
#pragma unroll
    for (int c = 0; c < 64; c++)
      val[c] = (int)p[c * 8] - 127;
  }

  else if constexpr (SRC == SRC_STEREO)
  { // 64 coefficient planes (simd_dct.cpp:1061-1099): coefficient c of stream position p at c * plane + p, so the
    // wave's 64 blocks are 64 consecutive bytes of every plane; bias +127 as in the q32 tier (:1020)
    // A wave-uniform base per plane (scalar adds) plus the lane's 32-bit offset: no per-lane 64-bit address arithmetic.
    const uint8_t *p = static_cast<const uint8_t *>(a.src) + blk0;
    const uint32_t l = valid ? lane : 0;
#pragma unroll
    for (int c = 0; c < 64; c++)
    { // readfirstlane pins the plane's base to scalar registers: global_load_ubyte v, v_lane, s[base]
      const uint64_t b = reinterpret_cast<uint64_t>(p + (size_t)c * a.pitch);
      typedef const uint8_t __attribute__((address_space(1))) *global_bytes; // keeps the load a global_load (not flat) after the integer round trip
      // (the builtin returns int: both halves go through uint32_t, or the low half's sign would smear over the high one)
      const uint32_t b_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(b >> 32)), b_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b);
      const global_bytes pc = reinterpret_cast<global_bytes>((uint64_t)b_hi << 32 | (uint64_t)b_lo);
      val[c] = (int)pc[l] - 127;
    }
  }
  else
  { // SRC_BLOCK (simd_dct.cpp:347-362): 64 contiguous bytes per block, coefficient (v,u) stored at u*8+v, bias +127/255*255
    const uint8_t *p = static_cast<const uint8_t *>(a.src) + (blk0 + (valid ? lane : 0)) * 64;
    uint32_t w[16];
#pragma unroll
    for (int j = 0; j < 4; j++)
    {
      const u32x4 q = *reinterpret_cast<const u32x4 *>(p + j * 16); // plain, not non-temporal: the four pieces of a block share a cache line
      w[4 * j] = q.x; w[4 * j + 1] = q.y; w[4 * j + 2] = q.z; w[4 * j + 3] = q.w;
    }
#pragma unroll
    for (int i = 0; i < 64; i++)
    {
      const int st = (i & 7) * 8 + (i >> 3); // natural v*8+u lives at stored u*8+v
      val[i] = (int)((w[st >> 2] >> (8 * (st & 3))) & 0xFF) - 127;
    }
  }

  scan_emit<RLE>(val, lv, rn, lane, nvalid, valid, blk0, a.levels, a.runs, a.counts);
}

struct SplitArgs
{
  const uint8_t *ycc;
  void *y, *cb, *cr;              // int16 planes (level-shifted) or 8-bit planes (k_split420<true>)
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

// one thread: 8 pixels x 2 rows (48 bytes in; 2 x 16 B of Y, 8 B of Cb, 8 B of Cr out).  U8_OUT: the planes stay 8-bit and unshifted
// (2 x 8 B of Y, 4 B of Cb, 4 B of Cr) -- what the 8-bit plane batches take (mdct_roundtrip_u8_batch, mdct_fwd_u8_i16_batch,
// mdct_fwd_quant32_u8_batch), which do the level shift themselves.
template <bool U8_OUT>
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
  constexpr int kShift = U8_OUT ? 0 : 128;
#define Y0(i) (byte_of<3 * (i)>(w0) - kShift)
#define Y1(i) (byte_of<3 * (i)>(w1) - kShift)
#define C4(k, i) (((byte_of<3 * (2 * (i)) + (k)>(w0) + byte_of<3 * (2 * (i) + 1) + (k)>(w0) + byte_of<3 * (2 * (i)) + (k)>(w1) + byte_of<3 * (2 * (i) + 1) + (k)>(w1) + 2) >> 2) - kShift)
  if constexpr (U8_OUT)
  {
    auto pack4 = [](int b0, int b1, int b2, int b3) { return (uint32_t)b0 | ((uint32_t)b1 << 8) | ((uint32_t)b2 << 16) | ((uint32_t)b3 << 24); };
    const u32x2_unaligned y0 = {pack4(Y0(0), Y0(1), Y0(2), Y0(3)), pack4(Y0(4), Y0(5), Y0(6), Y0(7))};
    const u32x2_unaligned y1 = {pack4(Y1(0), Y1(1), Y1(2), Y1(3)), pack4(Y1(4), Y1(5), Y1(6), Y1(7))};
    typedef unsigned int u32_unaligned __attribute__((aligned(1)));
    uint8_t *py = static_cast<uint8_t *>(a.y) + (size_t)(2 * rp) * a.pitch_y + (size_t)s * 8;
    __builtin_nontemporal_store(y0, reinterpret_cast<u32x2_unaligned *>(py));
    __builtin_nontemporal_store(y1, reinterpret_cast<u32x2_unaligned *>(py + a.pitch_y));
    __builtin_nontemporal_store(pack4(C4(1, 0), C4(1, 1), C4(1, 2), C4(1, 3)), reinterpret_cast<u32_unaligned *>(static_cast<uint8_t *>(a.cb) + (size_t)rp * a.pitch_c + (size_t)s * 4));
    __builtin_nontemporal_store(pack4(C4(2, 0), C4(2, 1), C4(2, 2), C4(2, 3)), reinterpret_cast<u32_unaligned *>(static_cast<uint8_t *>(a.cr) + (size_t)rp * a.pitch_c + (size_t)s * 4));
  }
  else
  {
    auto pack = [](int lo, int hi) { return ((uint32_t)lo & 0xFFFFu) | ((uint32_t)hi << 16); };
    const u32x4 y0 = {pack(Y0(0), Y0(1)), pack(Y0(2), Y0(3)), pack(Y0(4), Y0(5)), pack(Y0(6), Y0(7))};
    const u32x4 y1 = {pack(Y1(0), Y1(1)), pack(Y1(2), Y1(3)), pack(Y1(4), Y1(5)), pack(Y1(6), Y1(7))};
    const u32x2 cb = {pack(C4(1, 0), C4(1, 1)), pack(C4(1, 2), C4(1, 3))};
    const u32x2 cr = {pack(C4(2, 0), C4(2, 1)), pack(C4(2, 2), C4(2, 3))};
    int16_t *py = static_cast<int16_t *>(a.y) + (size_t)(2 * rp) * a.pitch_y + (size_t)s * 8;
    __builtin_nontemporal_store(y0, reinterpret_cast<u32x4 *>(py));
    __builtin_nontemporal_store(y1, reinterpret_cast<u32x4 *>(py + a.pitch_y));
    __builtin_nontemporal_store(cb, reinterpret_cast<u32x2 *>(static_cast<int16_t *>(a.cb) + (size_t)rp * a.pitch_c + (size_t)s * 4));
    __builtin_nontemporal_store(cr, reinterpret_cast<u32x2 *>(static_cast<int16_t *>(a.cr) + (size_t)rp * a.pitch_c + (size_t)s * 4));
  }
#undef Y0
#undef Y1
#undef C4
}


// ---------------------------------------------------------------------------------------
// k_huffman_rows: baseline Huffman coding (ITU-T T.81 Annex C, F.1.2) of the run/level records.
// One workgroup of 4 waves per block row; the row is one restart interval: DC predictor 0 at its
// start, byte-aligned, 1-padded segment at out + row * seg_stride, unstuffed (the container writer
// stuffs, simd_dct_amd/jfif.py).  The row is walked in chunks of 256 blocks, one block per lane, 64
// consecutive blocks per wave:
//   0. the head of the wave's records (the first kHuffStage pairs of each block) comes in with
//      8 + 4 B per lane loads -- the NEXT chunk's are issued before this one is coded -- and is parked
//      in LDS as one dword per pair, kRecSkew dwords per block, so that "entry i of every lane" hits
//      64 banks; a block with more pairs reads the rest from HBM;
//   1. DC value per lane, predecessor's by a wave shuffle (lane 0: prefetched with the records);
//   2. walk the pairs: each becomes its token (Huffman code and amplitude bits, <= 26 bits, and the
//      length) in place in LDS, the lengths are summed;   3. exclusive scan of the sums: shuffles in
//      the wave, 4 totals through LDS;
//   4. walk the tokens, shifting each into a 64-bit accumulator that is OR-ed, 32 bits at a time,
//      into an LDS ring of the row's bit stream (ds_or: the first and last word of a block are
//      shared with its neighbours);
//   5. the chunk's complete words leave, byte-swapped (the stream is MSB first), 4 B per lane, and
//      their ring slots are cleared; the trailing partial word simply stays in the ring.
// The ring holds kHuffRing words (4 bit/px over a chunk); a chunk with more bits than that is emitted
// in several windows (walk 4 repeated, each time keeping only the words of one window).
// ---------------------------------------------------------------------------------------
constexpr int kHuffWaves = 4;
constexpr int kHuffChunk = 64 * kHuffWaves; // blocks per chunk = lanes of the workgroup
constexpr int kHuffStage = 28;              // pairs per block parked in LDS, one dword each (a multiple of 4)
constexpr int kRecSkew = kHuffStage + 1;    // dwords per block in LDS (odd: entry i of 64 blocks = 64 banks)
constexpr int kHuffPieces = kHuffStage / 4; // a block's parked head in pieces of 4 pairs (8 B of levels + 4 B of runs)

__global__ __launch_bounds__(kHuffChunk) void k_huffman_rows(HuffArgs a)
{
  __shared__ uint32_t ac[256], dc[12];
  __shared__ uint32_t rec_all[kHuffWaves][64 * kRecSkew];
  __shared__ uint32_t ring[kHuffRing];
  __shared__ uint32_t tot[2][kHuffWaves];
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t row = a.by0 + blockIdx.x;
  ac[tid] = a.ac[tid];
  if (tid < 12)
    dc[tid] = a.dc[tid];
  for (uint32_t w = tid; w < kHuffRing; w += kHuffChunk)
    ring[w] = 0;
  const size_t row_blk0 = (size_t)row * a.bpr;
  const uint8_t *g_lv = reinterpret_cast<const uint8_t *>(a.levels) + row_blk0 * 128;
  const uint8_t *g_rn = a.runs + row_blk0 * 64;
  const uint8_t *g_ct = a.counts + row_blk0;
  uint32_t *rec_lds = rec_all[wave];
  HuffRowCoder<kHuffWaves, kHuffStage, true, false> coder;
  coder.ac = ac;
  coder.dc = dc;
  coder.ring = ring;
  coder.tot = tot;
  coder.dcx = nullptr;
  coder.out_w = reinterpret_cast<uint32_t *>(a.out + (size_t)row * a.seg_stride);
  coder.zrl = a.ac[0xF0]; // size << 16 | code (kernel arguments: scalar registers)
  coder.eob = a.ac[0x00];
  coder.bpr = a.bpr;

  // the head of the wave's 64 records: kHuffPieces pieces per block, kHuffPieces per lane
  u32x2 plv[kHuffPieces];
  uint32_t prn[kHuffPieces];
  int pn = 0;                          // this lane's pair count
  int pp_ct = 0, pp_rn = 1, pp_lv = 0; // head of the block before the wave's first one (its DC is lane 0's predictor)
  // Straight-line loads with clamped addresses: under a branch the compiler's wait-count bookkeeping serialises them
  // (one round trip per piece) and waits for the whole prefetch before the first walk.
  const uint32_t last_blk = a.bpr - 1;
  auto fetch = [&](uint32_t c0w) { // c0w: the wave's first block in the row
#pragma unroll
    for (int k = 0; k < kHuffPieces; k++)
    {
      const uint32_t q = k * 64 + lane, b = q / kHuffPieces, part = q - kHuffPieces * b, blk = min(c0w + b, last_blk);
      plv[k] = __builtin_nontemporal_load(reinterpret_cast<const u32x2 *>(g_lv + (blk * 128u + part * 8u))); // 32-bit offsets from a uniform base
      prn[k] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(g_rn + (blk * 64u + part * 4u)));
    }
    pn = g_ct[min(c0w + lane, last_blk)]; // lanes past the row's end are masked by `live`
    const uint32_t pb = min(max(c0w, 1u) - 1u, last_blk);
    pp_ct = g_ct[pb];
    pp_rn = g_rn[pb * 64u];
    pp_lv = *reinterpret_cast<const int16_t *>(g_lv + pb * 128u);
  };
  auto park = [&]() { // registers -> LDS, level and run of a pair in one dword
#pragma unroll
    for (int k = 0; k < kHuffPieces; k++)
    {
      const uint32_t q = k * 64 + lane, b = q / kHuffPieces, part = q - kHuffPieces * b;
      uint32_t *d = rec_lds + b * kRecSkew + part * 4;
      const uint32_t l[2] = {plv[k].x, plv[k].y};
#pragma unroll
      for (int e = 0; e < 4; e++)
        d[e] = ((l[e >> 1] >> (16 * (e & 1))) & 0xFFFFu) | (((prn[k] >> (8 * e)) & 0xFFu) << 16);
    }
  };
  auto wave_sync = [] {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };

  fetch(wave * 64);
  wg_sync(); // tables and the cleared ring
  for (uint32_t c0 = 0; c0 < a.bpr; c0 += kHuffChunk)
  {
    const uint32_t c0w = c0 + wave * 64, bx = c0w + lane;
    const bool live = bx < a.bpr;
    park();
    const int n = live ? min(pn, 64) : 0, prev_dc = (c0w > 0 && pp_ct > 0 && pp_rn == 0) ? pp_lv : 0;
    if (c0 + kHuffChunk < a.bpr)
      fetch(c0w + kHuffChunk); // in flight while this chunk is coded
    wave_sync();
    coder.chunk(c0, rec_lds + lane * kRecSkew, n, live, prev_dc, reinterpret_cast<const int16_t *>(g_lv + (live ? bx : 0u) * 128u), g_rn + (live ? bx : 0u) * 64u);
  }
  wg_sync();
  if (tid == 0)
    a.seg_bytes[row] = coder.finish();
}

// ---------------------------------------------------------------------------------------
// k_pack_*: the row segments of k_huffman_rows -> one contiguous, decodable scan (ITU-T T.81 B.1.1.5: a zero byte
// after every 0xFF of entropy-coded data; E.1.4 / B.2.1: RSTm, m = 0..7 cyclic, between the restart intervals).
// Three small launches: stuffed length of every row, exclusive scan over the rows, the copy.  The data is ~0.15 B/px.
// ---------------------------------------------------------------------------------------
struct PackArgs
{
  const uint8_t *seg;
  const uint32_t *seg_bytes;
  size_t seg_stride;
  uint8_t *out;
  unsigned long long capacity;
  unsigned long long *row_off; // [n_rows + 1]: k_pack_count leaves the lengths here, k_pack_scan turns them into offsets
  const uint32_t *ff_counts;   // not null: 0xFF bytes per row, counted by the producer of the segments (no k_pack_count pass)
  uint32_t n_rows, first_rst;
};

__global__ __launch_bounds__(256) void k_pack_count(PackArgs a)
{
  __shared__ uint32_t wave_tot[4];
  const uint32_t r = blockIdx.x, nb = (uint32_t)min((size_t)a.seg_bytes[r], a.seg_stride); // a length beyond the stride is not a row
  const uint32_t *p = reinterpret_cast<const uint32_t *>(a.seg + (size_t)r * a.seg_stride);
  uint32_t ff = 0;
  for (uint32_t i = threadIdx.x * 4; i < nb; i += 1024)
  {
    uint32_t m = ff_bytes(p[i >> 2]);
    if (nb - i < 4)
      m &= (1u << (8 * (nb - i))) - 1u; // the bytes past the row's end are not data
    ff += (uint32_t)__builtin_popcount(m);
  }
  uint32_t total;
  (void)wg_scan256(ff, wave_tot, total);
  if (threadIdx.x == 0)
    a.row_off[r] = (unsigned long long)nb + total + (r + 1 < a.n_rows ? 2u : 0u);
}

__global__ __launch_bounds__(256) void k_pack_scan(PackArgs a)
{ // one workgroup: lengths -> exclusive offsets, row_off[n_rows] = total
  __shared__ unsigned long long wave_tot[4];
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned long long carry = 0;
  for (uint32_t c0 = 0; c0 < a.n_rows; c0 += 256)
  {
    const uint32_t i = c0 + threadIdx.x;
    unsigned long long v = 0;
    if (i < a.n_rows)
      v = a.ff_counts ? (unsigned long long)min((size_t)a.seg_bytes[i], a.seg_stride) + a.ff_counts[i] + (i + 1 < a.n_rows ? 2u : 0u) : a.row_off[i];
    unsigned long long incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1)
    {
      const unsigned long long u = __shfl_up(incl, d, 64);
      if (lane >= (uint32_t)d)
        incl += u;
    }
    wg_sync();
    if (lane == 63)
      wave_tot[wave] = incl;
    wg_sync();
    unsigned long long before = 0, total = 0;
#pragma unroll
    for (uint32_t w = 0; w < 4; w++)
    {
      const unsigned long long t = wave_tot[w];
      before += w < wave ? t : 0ull;
      total += t;
    }
    if (i < a.n_rows)
      a.row_off[i] = carry + before + incl - v;
    carry += total;
  }
  if (threadIdx.x == 0)
    a.row_off[a.n_rows] = carry;
}

// stuffed length of row i in the scan, restart marker included (counted mode: the producer has counted the 0xFF bytes)
__device__ __forceinline__ unsigned long long pack_row_len(const PackArgs &a, uint32_t i)
{
  return (unsigned long long)min((size_t)a.seg_bytes[i], a.seg_stride) + a.ff_counts[i] + (i + 1 < a.n_rows ? 2u : 0u);
}

// One workgroup per row; the copy itself is pack_write_row (pack_rows.h).
// OWN_BASE (counted mode, few rows): the workgroup sums the lengths of the rows before it itself, so that no
// k_pack_scan launch stands between the coder and the copy; it also writes row_off[].
template <bool OWN_BASE>
__global__ __launch_bounds__(256) void k_pack_write(PackArgs a)
{
  __shared__ uint32_t wave_tot[4];
  __shared__ unsigned long long wave_sum[4];
  __shared__ __attribute__((aligned(16))) uint32_t stage[kPackStageWords];
  const uint32_t r = blockIdx.x, nb = (uint32_t)min((size_t)a.seg_bytes[r], a.seg_stride); // a length beyond the stride is not a row
  PackRow<false> row;
  row.begin(a.seg + (size_t)r * a.seg_stride, a.seg_stride, nb); // on its way while the row's place is worked out
  unsigned long long base, end;
  if (OWN_BASE)
  {
    unsigned long long s = 0;
    for (uint32_t i0 = threadIdx.x; i0 < r; i0 += 4 * 256)
    { // four rows per thread and step: eight independent loads in flight
      unsigned long long l[4];
#pragma unroll
      for (uint32_t k = 0; k < 4; k++)
        l[k] = i0 + 256 * k < r ? pack_row_len(a, i0 + 256 * k) : 0ull;
      s += (l[0] + l[1]) + (l[2] + l[3]);
    }
    base = wg_sum256(s, wave_sum);
    end = base + pack_row_len(a, r);
    if (threadIdx.x == 0)
    {
      a.row_off[r] = base;
      if (r + 1 == a.n_rows)
        a.row_off[a.n_rows] = end;
    }
  }
  else
  {
    base = a.row_off[r];
    end = a.row_off[r + 1];
  }
  if (end > a.capacity)
    return; // does not fit: the caller sees row_off[n_rows] > capacity
  row.finish(a.out + base, r + 1 < a.n_rows, (a.first_rst + r) & 7, stage, wave_tot);
}

// mdct_init: load this file's code object now rather than at the first stage call (mdct_kernels.hip: preload_kernels)
hipError_t preload_stage_kernels()
{
  hipFuncAttributes attr;
  const hipError_t e = hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&k_pack_count));
  if (e != hipSuccess)
    (void)hipGetLastError();
  return e;
}

} // namespace mdct

namespace
{

// ITU-T T.81 Annex K.3.3, Tables K.3-K.6: BITS and HUFFVAL of the typical Huffman tables
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

const uint8_t *huff_bits(int which) { return which == 0 ? kDcLumaBits : (which == 1 ? kAcLumaBits : (which == 2 ? kDcChromaBits : kAcChromaBits)); }
const uint8_t *huff_vals(int which) { return which == 1 ? kAcLumaVals : (which == 3 ? kAcChromaVals : kDcVals); }

// Annex C: codes in order of increasing length; entry = size << 16 | code
void huff_build(int which, uint32_t *tab, int ntab)
{
  const uint8_t *bits = huff_bits(which), *vals = huff_vals(which);
  for (int i = 0; i < ntab; i++)
    tab[i] = 0;
  uint32_t code = 0;
  int k = 0;
  for (int len = 1; len <= 16; len++)
  {
    for (int i = 0; i < bits[len - 1]; i++, k++)
      if (vals[k] < ntab)
        tab[vals[k]] = ((uint32_t)len << 16) | code++;
    code <<= 1;
  }
}

} // namespace
// for the fused pixels -> Huffman rows entry points (mdct_api.hip)
extern "C" __attribute__((visibility("hidden"))) void mdct_huff_build(int which, uint32_t *tab, int ntab) { huff_build(which, tab, ntab); }
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
  hipStream_t s = (hipStream_t)stream;
#define MDCT_SCAN(SRC) \
  do \
  { \
    const dim3 g((uint32_t)((n + mdct::scan_wg(SRC) - 1) / mdct::scan_wg(SRC))), b(mdct::scan_wg(SRC)); \
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

int mdct_huffman_spec(int which, uint8_t *bits16, uint8_t *vals, int *nvals)
{
  if (which < 0 || which > 3 || !bits16 || !vals || !nvals)
    return mdct_set_error(MDCT_INVALID_PARAMETER, "table 0..3 (DC luma, AC luma, DC chroma, AC chroma) and non-null outputs");
  int n = 0;
  for (int i = 0; i < 16; i++)
  {
    bits16[i] = huff_bits(which)[i];
    n += bits16[i];
  }
  for (int i = 0; i < n; i++)
    vals[i] = huff_vals(which)[i];
  *nvals = n;
  return MDCT_SUCCESS;
}

size_t mdct_huffman_seg_stride(size_t sizeX) { return (sizeX / 8) * 208 + 8; }

int mdct_huffman_rows(const int16_t *levels, const uint8_t *runs, const uint8_t *counts, size_t sizeX, size_t sizeY, size_t by0, size_t by1, int chroma, uint8_t *out, size_t seg_stride,
                      uint32_t *seg_bytes, void *stream)
{
  if (!levels || !runs || !counts || !out || !seg_bytes)
    return mdct_set_error(MDCT_INVALID_PARAMETER, "null pointer");
  if (sizeX == 0 || sizeX % 8 != 0 || sizeY % 8 != 0)
    return mdct_set_error(MDCT_NOT_SUPPORTED, "plane %zux%zu is not a multiple of 8x8", sizeX, sizeY);
  const size_t bpr = sizeX / 8;
  if (by0 > by1 || by1 > sizeY / 8 || bpr > 0xFFFF)
    return mdct_set_error(MDCT_INVALID_PARAMETER, "bad block-row range [%zu,%zu) for %zu rows, or more than 65535 blocks per row (a restart interval)", by0, by1, sizeY / 8);
  if (seg_stride < bpr * 208 + 8 || seg_stride % 4 != 0 || ((uintptr_t)out & 3))
    return mdct_set_error(MDCT_INVALID_PARAMETER, "seg_stride must be a multiple of 4 and >= 208 * blocks per row + 8 = %zu (worst case of F.1.2); out 4-byte aligned", bpr * 208 + 8);
  if (by0 == by1)
    return MDCT_SUCCESS;
  mdct::HuffArgs a;
  a.levels = levels;
  a.runs = runs;
  a.counts = counts;
  a.out = out;
  a.seg_bytes = seg_bytes;
  a.seg_stride = seg_stride;
  a.bpr = (uint32_t)bpr;
  a.by0 = (uint32_t)by0;
  huff_build(chroma ? 2 : 0, a.dc, 12);
  huff_build(chroma ? 3 : 1, a.ac, 256);
  hipLaunchKernelGGL(mdct::k_huffman_rows, dim3((uint32_t)(by1 - by0)), dim3(mdct::kHuffChunk), 0, (hipStream_t)stream, a);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? MDCT_SUCCESS : mdct_set_error(MDCT_NOT_SUPPORTED, "huffman kernel launch: %s", hipGetErrorString(e));
}

static int pack_rows(const uint8_t *segments, const uint32_t *seg_bytes, const uint32_t *ff_counts, size_t seg_stride, size_t n_rows, int first_rst, uint8_t *out, size_t out_capacity,
                     uint64_t *row_offsets, void *stream);

int mdct_jpeg_pack_rows(const uint8_t *segments, const uint32_t *seg_bytes, size_t seg_stride, size_t n_rows, int first_rst, uint8_t *out, size_t out_capacity, uint64_t *row_offsets,
                        void *stream)
{
  return pack_rows(segments, seg_bytes, nullptr, seg_stride, n_rows, first_rst, out, out_capacity, row_offsets, stream);
}

int mdct_jpeg_pack_rows_counted(const uint8_t *segments, const uint32_t *seg_bytes, const uint32_t *ff_counts, size_t seg_stride, size_t n_rows, int first_rst, uint8_t *out,
                                size_t out_capacity, uint64_t *row_offsets, void *stream)
{
  if (!ff_counts)
    return mdct_set_error(MDCT_INVALID_PARAMETER, "null pointer");
  return pack_rows(segments, seg_bytes, ff_counts, seg_stride, n_rows, first_rst, out, out_capacity, row_offsets, stream);
}

static int pack_rows(const uint8_t *segments, const uint32_t *seg_bytes, const uint32_t *ff_counts, size_t seg_stride, size_t n_rows, int first_rst, uint8_t *out, size_t out_capacity,
                     uint64_t *row_offsets, void *stream)
{
  if (!segments || !seg_bytes || !out || !row_offsets)
    return mdct_set_error(MDCT_INVALID_PARAMETER, "null pointer");
  if (seg_stride % 4 != 0 || ((uintptr_t)segments & 3) || ((uintptr_t)row_offsets & 7))
    return mdct_set_error(MDCT_INVALID_PARAMETER, "segments 4-byte aligned with a stride that is a multiple of 4; row_offsets 8-byte aligned");
  if (n_rows > 0x7FFFFFFFull || first_rst < 0 || first_rst > 7)
    return mdct_set_error(MDCT_INVALID_PARAMETER, "more than 2^31 rows, or first_rst outside 0..7");
  mdct::PackArgs a;
  a.seg = segments;
  a.seg_bytes = seg_bytes;
  a.seg_stride = seg_stride;
  a.out = out;
  a.capacity = out_capacity;
  a.row_off = reinterpret_cast<unsigned long long *>(row_offsets);
  a.n_rows = (uint32_t)n_rows;
  a.first_rst = (uint32_t)first_rst;
  a.ff_counts = ff_counts;
  hipStream_t s = (hipStream_t)stream;
  // counted rows, and few enough of them that every workgroup can sum the lengths before its own: one launch
  static const bool never_own = getenv("MDCT_PACK_SCAN_KERNEL") != nullptr;
  if (n_rows && ff_counts && n_rows <= 16384 && !never_own)
    hipLaunchKernelGGL(mdct::k_pack_write<true>, dim3(a.n_rows), dim3(256), 0, s, a);
  else
  {
    if (n_rows && !ff_counts)
      hipLaunchKernelGGL(mdct::k_pack_count, dim3(a.n_rows), dim3(256), 0, s, a);
    hipLaunchKernelGGL(mdct::k_pack_scan, dim3(1), dim3(256), 0, s, a);
    if (n_rows)
      hipLaunchKernelGGL(mdct::k_pack_write<false>, dim3(a.n_rows), dim3(256), 0, s, a);
  }
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? MDCT_SUCCESS : mdct_set_error(MDCT_NOT_SUPPORTED, "pack kernel launch: %s", hipGetErrorString(e));
}

static int run_split420(const uint8_t *ycc, size_t pitch, size_t sizeX, size_t sizeY, void *y, void *cb, void *cr, size_t pitch_y, size_t pitch_c, bool u8_out, void *stream)
{
  if (ycc == nullptr || y == nullptr || cb == nullptr || cr == nullptr)
    return mdct_set_error(MDCT_INVALID_PARAMETER, "null pointer");
  if (sizeX == 0 || sizeX % 16 != 0 || sizeY % 16 != 0) // both chroma planes must come out as whole 8x8 blocks
    return mdct_set_error(MDCT_NOT_SUPPORTED, "image %zux%zu is not a multiple of 16x16", sizeX, sizeY);
  if (pitch < 3 * sizeX || pitch_y < sizeX || pitch_c < sizeX / 2)
    return mdct_set_error(MDCT_INVALID_PARAMETER, "pitch smaller than a row");
  if (!u8_out && ((((uintptr_t)y | (pitch_y * 2)) & 15) || (((uintptr_t)cb | (uintptr_t)cr | (pitch_c * 2)) & 7)))
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
  const dim3 grid((uint32_t)((n + mdct::kWG - 1) / mdct::kWG));
  if (u8_out)
    hipLaunchKernelGGL(mdct::k_split420<true>, grid, dim3(mdct::kWG), 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(mdct::k_split420<false>, grid, dim3(mdct::kWG), 0, (hipStream_t)stream, a);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? MDCT_SUCCESS : mdct_set_error(MDCT_NOT_SUPPORTED, "split kernel launch: %s", hipGetErrorString(e));
}

int mdct_split420_u8(const uint8_t *ycc, size_t pitch, size_t sizeX, size_t sizeY, int16_t *y, int16_t *cb, int16_t *cr, size_t pitch_y, size_t pitch_c, void *stream)
{
  return run_split420(ycc, pitch, sizeX, sizeY, y, cb, cr, pitch_y, pitch_c, false, stream);
}

int mdct_split420_u8_planes(const uint8_t *ycc, size_t pitch, size_t sizeX, size_t sizeY, uint8_t *y, uint8_t *cb, uint8_t *cr, size_t pitch_y, size_t pitch_c, void *stream)
{
  return run_split420(ycc, pitch, sizeX, sizeY, y, cb, cr, pitch_y, pitch_c, true, stream);
}

} // extern "C"
