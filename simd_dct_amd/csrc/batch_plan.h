// batch_plan.h -- host-only half of the plane-batch entry points (mdct_*_i16_batch, mdct_batch_*): how a list of
// separately allocated planes becomes the 1-D grid of 64-block tiles k_i16_batch runs on (mdct_kernels.hip).
// Nothing of the HIP runtime in here: tests/batch_plan_driver.cpp builds it with plain g++ (sanitizers on) and walks
// every tile index of a launch through batch_locate(), the arithmetic the kernel uses, to see that each tile of each
// plane is reached exactly once.
//
// The reference's only batching affordance is the caller-side row range of every variant (simd_dct.cpp:2243-2261);
// BASELINE.json's configs[2] (Y + Cb + Cr, own tables) and configs[3] (256 independent planes) need one of the engine's own.
#ifndef MDCT_BATCH_PLAN_H
#define MDCT_BATCH_PLAN_H

#include <stddef.h>
#include <stdint.h>

#include <vector>

#include "mdct.h"

// The index arithmetic below is ONE piece of code for the host walk (tests/batch_plan_driver.cpp, plain g++ with sanitizers) and
// for the kernels (mdct_kernels.hip: k_i16_batch, k_u8_batch call batch_plane_of / magic_apply themselves).
#if defined(__HIPCC__)
#define MDCT_HD __host__ __device__ __forceinline__
#else
#define MDCT_HD inline
#endif

namespace mdct
{

// What a wave needs of its plane: one 64-byte descriptor, read with scalar loads (one s_load_dwordx16).
struct BatchDesc
{
  const void *from;             // int16 planes (k_i16_batch) or 8-bit planes (k_u8_batch)
  void *to;
  uint64_t pitch_in, pitch_out; // elements (int16 planes) / bytes (8-bit planes)
  uint32_t bpr;                 // blocks per block row (sizeX / 8)
  uint32_t tiles;               // 64-block tiles per UNIT, the last one possibly partial: unit = one block row, (bpr + 63) / 64 tiles;
                                // kDescPaired: unit = a PAIR of block rows tiled as one run of 2 * bpr blocks, bpr / 32 whole tiles
  uint32_t tiles_m, tiles_s;    // exact division by `tiles` (MagicDiv: multiplier, shifts sh1 | sh2 << 8)
  uint32_t first;               // index of the plane's first tile in the launch
  uint32_t table;               // byte offset of the plane's tables from the table base
  uint32_t has_lut;             // bit 0: fused round trip quantises / dequantises in between; bit 1: kDescPaired
  uint32_t rows;                // block rows (the kernels read it for paired planes only: an odd last row has no partner)
};
static_assert(sizeof(BatchDesc) == 64, "one s_load_dwordx16");
// Paired rows (round 6): a plane whose rows end in HALF a tile (bpr % 64 == 32: the 3840-wide chroma planes of an 8K 4:2:0 frame, 7.5 tiles
// per row) spends every 8th wave on 32 blocks.  The kernels whose lanes address their blocks independently (k_u8_batch, k_q32_batch) tile
// such a plane over pairs of block rows instead: 2 * bpr blocks = bpr / 32 whole tiles, the middle one straddling the row boundary with its
// lanes 0..31 at the end of the upper row and 32..63 at the start of the lower one -- 12,150 waves instead of 12,420 for that frame.
constexpr uint32_t kDescLut = 1, kDescPaired = 2;

// q = n / d for every 32-bit n without a division (Granlund & Montgomery 1994, Figure 4.1):
// t = mulhi(n, m); q = (t + ((n - t) >> sh1)) >> sh2.  d = 1: m = 0, both shifts 0.
struct MagicDiv
{
  uint32_t m, s; // s = sh1 | sh2 << 8
};
inline MagicDiv magic_div(uint32_t d)
{
  if (d <= 1)
    return MagicDiv{0, 0};
  uint32_t l = 0;
  while ((1ull << l) < d)
    l++;
  const uint64_t m = ((1ull << 32) * ((1ull << l) - d)) / d + 1;
  return MagicDiv{(uint32_t)m, 1u | ((l - 1) << 8)};
}
MDCT_HD uint32_t magic_apply(uint32_t n, uint32_t m, uint32_t s)
{
  const uint32_t t = (uint32_t)(((uint64_t)n * m) >> 32); // device: one s_mul_hi_u32 / v_mul_hi_u32
  return (t + ((n - t) >> (s & 0xFF))) >> (s >> 8);
}

constexpr int kBatchChain = 8;                // up to this many planes of different shapes are told apart by a compare chain
// Grid limit of one launch: a tile is a 64-thread workgroup and the HIP runtime refuses launches of more than 2^32 - 1 threads
// (gridDim.x * blockDim.x), so 2^26 - 1 tiles (= 4.29e9 blocks, 275 Gpx); longer lists are split into several launches.
constexpr uint32_t kBatchMaxTiles = 0xFFFFFFFFu / 64;

// Which plane tile `w` of a launch belongs to -- the ONE implementation, run by the kernels and walked by the CPU test:
// equal tile grids: a magic multiply; up to kBatchChain different ones: a compare chain on the firsts the header carries
// (UINT32_MAX beyond n); more: the last plane whose first tile is <= w, `first_of(k)` reading descriptor k's `first`.
template <class FirstOf>
MDCT_HD uint32_t batch_plane_of(uint32_t w, uint32_t n, uint32_t uniform, uint32_t pp_m, uint32_t pp_s, const uint32_t (&first8)[kBatchChain], FirstOf first_of)
{
  if (uniform)
    return magic_apply(w, pp_m, pp_s);
  uint32_t p = 0;
  for (int i = 1; i < kBatchChain; i++)
    p += w >= first8[i] ? 1u : 0u;
  if (n > (uint32_t)kBatchChain)
  {
    uint32_t lo = 0, hi = n;
    while (hi - lo > 1)
    {
      const uint32_t mid = (lo + hi) >> 1;
      if (first_of(mid) <= w)
        lo = mid;
      else
        hi = mid;
    }
    p = lo;
  }
  return p;
}

struct BatchLayout
{
  std::vector<BatchDesc> descs;  // one per non-empty plane, in order
  std::vector<int> plane;        // descs[k] describes planes[plane[k]]
  std::vector<int> tables;       // local table slot -> the caller's table id
  uint32_t total = 0;            // tiles in the launch
  uint32_t uniform = 0;          // every plane has the same tile grid
  uint32_t per_plane = 0;        // tiles per plane when uniform
  MagicDiv pp{0, 0};
  uint32_t first8[kBatchChain];
  int with_lut = 0;              // planes whose has_lut is set
  int consumed = 0;              // planes taken from the list (empty ones included)
};

// tiles a plane takes in a launch
inline uint64_t batch_tiles_of(uint64_t bpr, uint64_t rows, bool paired)
{
  return paired ? (rows / 2) * (bpr / 32) + (rows & 1) * ((bpr + 63) / 64) : ((bpr + 63) / 64) * rows;
}

// Where local tile `lt` of a plane lies: its block row, its first block within that row, how many of its 64 blocks lie in that row, and
// whether the rest (lanes >= s) continue at block 0 of the next row (a paired plane's middle tile).  ONE implementation: the kernels run
// it on the scalar unit (mdct_kernels.hip: batch_tile), tests/batch_plan_driver.cpp walks it under the sanitizers.
struct BatchPos
{
  uint32_t row, b0, s, straddle;
};
MDCT_HD BatchPos batch_pos(uint32_t lt, uint32_t bpr, uint32_t tiles, uint32_t tiles_m, uint32_t tiles_s, uint32_t flags, uint32_t rows)
{
  const uint32_t u = magic_apply(lt, tiles_m, tiles_s), blk0 = (lt - u * tiles) * 64;
  BatchPos p;
  if (!(flags & kDescPaired))
  {
    p.row = u;
    p.b0 = blk0;
    p.s = bpr - blk0 < 64u ? bpr - blk0 : 64u;
    p.straddle = 0;
    return p;
  }
  const uint32_t second = blk0 >= bpr ? 1u : 0u;
  p.row = 2 * u + second;
  p.b0 = second ? blk0 - bpr : blk0;
  p.s = bpr - p.b0 < 64u ? bpr - p.b0 : 64u;
  p.straddle = (!second && p.s < 64u && p.row + 1 < rows) ? 1u : 0u;
  return p;
}

// Lays out planes [i0, n) greedily: as many as fit `blob_bytes` of tables (table_size bytes each) + descriptors
// (blob_bytes = 0: no limit) and the grid limit.  table_id[i] < 0: plane i needs no table slot; equal ids share a slot.
// has_lut[i] goes into the descriptor.  Planes without blocks take no descriptor.  Always consumes at least one plane
// when i0 < n (a single plane that exceeds the grid limit yields consumed = 0: the caller reports it).
template <class Plane> // mdct_plane_i16 or mdct_plane_u8: the same field names, pitches in the plane's own unit
// pair_rows: planes with bpr % 64 == 32 are tiled over pairs of block rows (kDescPaired) -- only for kernels that understand it.
inline void batch_layout(const Plane *planes, const int *table_id, const unsigned char *has_lut, int i0, int n, size_t blob_bytes, size_t table_size, BatchLayout &out,
                         bool pair_rows = false)
{
  out = BatchLayout();
  uint64_t run = 0;
  for (int i = i0; i < n; i++)
  {
    const Plane &p = planes[i];
    const uint64_t bpr = p.sizeX / 8, rows = p.sizeY / 8;
    if (bpr == 0 || rows == 0)
    {
      out.consumed++;
      continue;
    }
    // (paired: the lane offsets of a straddling tile are 32-bit byte offsets from the tile's base -- pitches far below 4 GiB / 16 only)
    const bool paired = pair_rows && bpr % 64 == 32 && rows >= 2 && (uint64_t)p.pitch_in < (1ull << 26) && (uint64_t)p.pitch_out < (1ull << 26);
    const uint64_t tiles = paired ? bpr / 32 : (bpr + 63) / 64;
    if (bpr > 0xFFFFFFFFull || rows > 0xFFFFFFFFull || ((bpr + 63) / 64) * rows / rows != (bpr + 63) / 64)
      break;
    const uint64_t mine = batch_tiles_of(bpr, rows, paired);
    if (run + mine > kBatchMaxTiles)
      break;
    int slot = -1;
    bool new_table = false;
    if (table_id[i] >= 0)
    {
      for (size_t k = 0; k < out.tables.size() && slot < 0; k++)
        if (out.tables[k] == table_id[i])
          slot = (int)k;
      new_table = slot < 0;
      if (new_table)
        slot = (int)out.tables.size();
    }
    if (blob_bytes && (out.tables.size() + (new_table ? 1 : 0)) * table_size + (out.descs.size() + 1) * sizeof(BatchDesc) > blob_bytes)
      break;
    if (new_table)
      out.tables.push_back(table_id[i]);
    BatchDesc d;
    d.from = p.from;
    d.to = p.to;
    d.pitch_in = p.pitch_in;
    d.pitch_out = p.pitch_out;
    d.bpr = (uint32_t)bpr;
    d.tiles = (uint32_t)tiles;
    const MagicDiv md = magic_div(d.tiles);
    d.tiles_m = md.m;
    d.tiles_s = md.s;
    d.first = (uint32_t)run;
    d.table = slot < 0 ? 0u : (uint32_t)(slot * table_size);
    d.has_lut = (has_lut[i] ? kDescLut : 0u) | (paired ? kDescPaired : 0u);
    d.rows = (uint32_t)rows;
    out.with_lut += has_lut[i] ? 1 : 0;
    out.descs.push_back(d);
    out.plane.push_back(i);
    out.consumed++;
    run += mine;
  }
  out.total = (uint32_t)run;
  const size_t nd = out.descs.size();
  out.uniform = nd > 0;
  auto tiles_of = [&](size_t k) { return batch_tiles_of(out.descs[k].bpr, out.descs[k].rows, (out.descs[k].has_lut & kDescPaired) != 0); };
  for (size_t k = 1; k < nd; k++)
    if (tiles_of(k) != tiles_of(0))
      out.uniform = 0;
  out.per_plane = out.uniform ? (uint32_t)tiles_of(0) : 0;
  out.pp = magic_div(out.per_plane);
  for (int k = 0; k < kBatchChain; k++)
    out.first8[k] = (size_t)k < nd ? out.descs[k].first : 0xFFFFFFFFu;
}

// Where tile `w` of a launch lies, as the kernels compute it (batch_plane_of, then the row / tile split by the plane's MagicDiv).
struct BatchWhere
{
  uint32_t p, row, tile; // tile = b0 / 64 for planes that are not paired
  BatchPos pos;
};
inline BatchWhere batch_locate(const BatchDesc *descs, uint32_t n, uint32_t uniform, MagicDiv pp, const uint32_t (&first8)[kBatchChain], uint32_t w)
{
  const uint32_t p = batch_plane_of(w, n, uniform, pp.m, pp.s, first8, [descs](uint32_t k) { return descs[k].first; });
  const BatchDesc &d = descs[p];
  const BatchPos pos = batch_pos(w - d.first, d.bpr, d.tiles, d.tiles_m, d.tiles_s, d.has_lut, d.rows);
  return BatchWhere{p, pos.row, pos.b0 / 64, pos};
}

} // namespace mdct
#endif
