// batch_plan_driver.cpp -- TEST INFRASTRUCTURE: the host-only half of the plane-batch entry points (csrc/batch_plan.h)
// built with plain g++ (ASan + UBSan) by tests/test_batch.py.  Walks every tile index of a laid-out launch through
// batch_locate() -- the arithmetic k_i16_batch runs on the GPU -- and checks that each tile of each plane is reached
// exactly once, for equal shapes (magic multiply), up to 8 different shapes (compare chain) and more (binary search);
// the division-free quotient against `/` on boundary and random operands; the greedy chunking against the argument-block
// budget, table sharing, empty planes and the grid limit.
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "batch_plan.h"

using namespace mdct;

#define CHECK(c)                                                              \
  do                                                                          \
  {                                                                           \
    if (!(c))                                                                 \
    {                                                                         \
      fprintf(stderr, "%s:%d: CHECK failed: %s\n", __FILE__, __LINE__, #c);   \
      exit(1);                                                                \
    }                                                                         \
  } while (0)

static void check_magic()
{
  std::mt19937_64 rng(7);
  std::vector<uint32_t> ds;
  for (uint32_t d = 1; d <= 5000; d++)
    ds.push_back(d);
  for (int k = 1; k < 32; k++)
    for (int o = -2; o <= 2; o++)
      ds.push_back((uint32_t)((1ll << k) + o));
  ds.push_back(0xFFFFFFFFu);
  ds.push_back(0x7FFFFFFFu);
  for (int i = 0; i < 20000; i++)
    ds.push_back((uint32_t)(rng() >> (rng() % 32)) | 1u);
  for (uint32_t d : ds)
  {
    if (d == 0)
      continue;
    const MagicDiv k = magic_div(d);
    const uint32_t ns[] = {0u, 1u, d - 1, d, d + 1, 2 * d - 1, 2 * d, 0xFFFFFFFFu, 0xFFFFFFFEu, 0x80000000u, 0x7FFFFFFFu, (uint32_t)rng(), (uint32_t)rng(), (uint32_t)rng(),
                           (0xFFFFFFFFu / d) * d, (0xFFFFFFFFu / d) * d - 1};
    for (uint32_t n : ns)
      CHECK(magic_apply(n, k.m, k.s) == n / d);
  }
}

struct Shape
{
  size_t w, h;
};

static std::vector<mdct_plane_i16> make_planes(const std::vector<Shape> &shapes)
{
  std::vector<mdct_plane_i16> v(shapes.size());
  for (size_t i = 0; i < shapes.size(); i++)
  {
    v[i].from = reinterpret_cast<const int16_t *>(0x1000000 * (i + 1));
    v[i].to = reinterpret_cast<int16_t *>(0x1000000 * (i + 1) + 0x800000);
    v[i].pitch_in = shapes[i].w + 8;
    v[i].pitch_out = shapes[i].w + 16;
    v[i].sizeX = shapes[i].w;
    v[i].sizeY = shapes[i].h;
    v[i].lut = nullptr;
  }
  return v;
}

// every BLOCK of every plane by exactly one lane of exactly one tile, in plane order; returns the launch's tile count.
// (block coverage rather than tile coverage: a paired plane's middle tile covers blocks of two rows.)
template <class Plane>
static uint32_t walk(const BatchLayout &lay, const std::vector<Plane> &planes)
{
  std::vector<std::vector<unsigned char>> seen(lay.descs.size());
  for (size_t k = 0; k < lay.descs.size(); k++)
    seen[k].assign((size_t)lay.descs[k].bpr * lay.descs[k].rows, 0);
  uint32_t last_p = 0;
  for (uint32_t w = 0; w < lay.total; w++)
  {
    const BatchWhere at = batch_locate(lay.descs.data(), (uint32_t)lay.descs.size(), lay.uniform, lay.pp, lay.first8, w);
    CHECK(at.p < lay.descs.size() && at.p >= last_p);
    last_p = at.p;
    const BatchDesc &d = lay.descs[at.p];
    const bool paired = (d.has_lut & kDescPaired) != 0;
    CHECK(at.row < d.rows && at.pos.b0 < d.bpr && at.pos.s >= 1 && at.pos.s <= 64 && at.pos.b0 + at.pos.s <= d.bpr);
    CHECK(at.pos.b0 % 64 == 0 || paired);
    CHECK(!at.pos.straddle || (paired && at.pos.s % 16 == 0 && at.pos.b0 + at.pos.s == d.bpr && at.row + 1 < d.rows)); // what k_q32_batch's stores rely on
    CHECK(at.pos.s == 64 || at.pos.b0 + at.pos.s == d.bpr);
    for (uint32_t lane = 0; lane < (at.pos.straddle ? 64u : at.pos.s); lane++)
    { // the lane's block as the kernels address it
      const uint32_t row = lane < at.pos.s ? at.row : at.row + 1, bx = lane < at.pos.s ? at.pos.b0 + lane : lane - at.pos.s;
      CHECK(row < d.rows && bx < d.bpr);
      unsigned char &s = seen[at.p][(size_t)row * d.bpr + bx];
      CHECK(s == 0);
      s = 1;
    }
    const Plane &pl = planes[lay.plane[at.p]];
    CHECK(d.from == pl.from && d.to == pl.to && d.pitch_in == pl.pitch_in && d.pitch_out == pl.pitch_out && d.bpr == pl.sizeX / 8 && d.rows == pl.sizeY / 8);
    CHECK(d.tiles == (paired ? d.bpr / 32 : (d.bpr + 63) / 64));
  }
  for (auto &v : seen)
    for (unsigned char s : v)
      CHECK(s == 1);
  return lay.total;
}

int main()
{
  check_magic();

  { // BASELINE.json configs[2]: Y 7680x4320 + Cb/Cr 3840x2160, a table each (two of them equal)
    auto planes = make_planes({{7680, 4320}, {3840, 2160}, {3840, 2160}});
    const int ids[] = {0, 1, 1};
    const unsigned char has[] = {1, 1, 1};
    BatchLayout lay;
    batch_layout(planes.data(), ids, has, 0, 3, 3584, 512, lay);
    CHECK(lay.consumed == 3 && lay.descs.size() == 3 && lay.tables.size() == 2 && !lay.uniform && lay.with_lut == 3);
    CHECK(lay.descs[1].table == 512 && lay.descs[2].table == 512 && lay.descs[0].table == 0);
    CHECK(lay.descs[1].tiles == 8 && lay.descs[1].bpr == 480); // 7.5 tiles: the last one is partial
    CHECK(walk(lay, planes) == 15u * 540 + 2u * 8 * 270);
  }
  { // configs[3] in miniature: equal shapes take the magic multiply; more planes than one argument block holds
    std::vector<Shape> shapes(200, Shape{520, 72}); // 65 blocks per row: 2 tiles, the second holds one block
    auto planes = make_planes(shapes);
    std::vector<int> ids(200, 0);
    std::vector<unsigned char> has(200, 0);
    int i0 = 0, launches = 0;
    while (i0 < 200)
    {
      BatchLayout lay;
      batch_layout(planes.data(), ids.data(), has.data(), i0, 200, 3584, 512, lay);
      CHECK(lay.consumed > 0 && lay.uniform && lay.per_plane == 18 && lay.tables.size() == 1);
      CHECK(lay.tables.size() * 512 + lay.descs.size() * 64 <= 3584);
      CHECK(lay.consumed == 48 || i0 + lay.consumed == 200); // (3584 - 512) / 64
      CHECK(lay.plane.front() == i0);
      walk(lay, planes);
      i0 += lay.consumed;
      launches++;
    }
    CHECK(launches == 5);
    BatchLayout all; // device-memory form: no budget, one launch
    batch_layout(planes.data(), ids.data(), has.data(), 0, 200, 0, 512, all);
    CHECK(all.consumed == 200 && all.total == 200u * 18);
    walk(all, planes);
  }
  { // many different shapes: chain (<= 8) and binary search (> 8), empty planes in between, distinct and shared tables
    std::mt19937 rng(3);
    for (int n : {1, 2, 7, 8, 9, 16, 33})
    {
      std::vector<Shape> shapes;
      std::vector<int> ids;
      std::vector<unsigned char> has;
      for (int i = 0; i < n; i++)
      {
        shapes.push_back(Shape{(size_t)(8 * (1 + rng() % 300)), (size_t)(8 * (1 + rng() % 40))});
        ids.push_back(i % 3 == 2 ? -1 : (int)(rng() % 4));
        has.push_back(ids.back() >= 0);
        if (i % 5 == 4)
        {
          shapes.push_back(Shape{0, 64});
          ids.push_back(0);
          has.push_back(1);
        }
      }
      auto planes = make_planes(shapes);
      BatchLayout lay;
      batch_layout(planes.data(), ids.data(), has.data(), 0, (int)planes.size(), 0, 512, lay);
      CHECK(lay.consumed == (int)planes.size() && (int)lay.descs.size() == n && lay.tables.size() <= 4);
      walk(lay, planes);
      for (size_t k = 0; k < lay.descs.size(); k++)
      {
        const int id = ids[lay.plane[k]];
        CHECK(id < 0 ? lay.descs[k].table == 0 && !lay.descs[k].has_lut : lay.tables[lay.descs[k].table / 512] == id);
      }
    }
  }
  { // the grid limit (kBatchMaxTiles = 2^26 - 1: 64-thread tiles, < 2^32 threads per launch): planes are taken while the launch stays
    // below it; one plane beyond it is refused
    static_assert(kBatchMaxTiles == 67108863u, "2^26 - 1");
    auto planes = make_planes({{8 * 64 * 4000, 8 * 10000}, {8 * 64 * 4000, 8 * 10000}, {8 * 64 * 7000, 8 * 10000}});
    const int ids[] = {0, 0, 0};
    const unsigned char has[] = {1, 1, 1};
    BatchLayout lay;
    batch_layout(planes.data(), ids, has, 0, 3, 0, 512, lay);
    CHECK(lay.consumed == 1 && lay.total == 40000000u);
    batch_layout(planes.data(), ids, has, 1, 3, 0, 512, lay);
    CHECK(lay.consumed == 1);
    batch_layout(planes.data(), ids, has, 2, 3, 0, 512, lay);
    CHECK(lay.consumed == 0 && lay.descs.empty());
    // exactly at the limit is taken, one tile more is not
    auto edge = make_planes({{8 * 64 * 8191, 8 * 8193}, {8 * 64, 8 * 1}});
    CHECK(8191ull * 8193ull == 67108863ull);
    batch_layout(edge.data(), ids, has, 0, 2, 0, 512, lay);
    CHECK(lay.consumed == 1 && lay.total == kBatchMaxTiles);
  }
  { // 8-bit planes lay out through the same template (pitches are the plane's own unit: bytes)
    std::vector<mdct_plane_u8> planes(3);
    const size_t w[3] = {7680, 3840, 3840}, h[3] = {4320, 2160, 2160};
    for (int i = 0; i < 3; i++)
      planes[i] = mdct_plane_u8{reinterpret_cast<const uint8_t *>((uintptr_t)0x1000 * (i + 1)), reinterpret_cast<uint8_t *>((uintptr_t)0x100000 * (i + 1)), w[i] + 16, w[i], w[i], h[i], nullptr};
    const int ids[] = {0, 1, 1};
    const unsigned char has[] = {1, 1, 1};
    BatchLayout lay;
    batch_layout(planes.data(), ids, has, 0, 3, 3584, 512, lay);
    CHECK(lay.consumed == 3 && lay.descs.size() == 3 && lay.tables.size() == 2 && lay.total == 15u * 540 + 2 * 8u * 270);
    CHECK(lay.descs[0].pitch_in == 7696 && lay.descs[1].first == 8100 && lay.descs[2].first == 8100 + 2160 && lay.descs[2].table == 512);
    for (uint32_t w0 : {0u, 8099u, 8100u, 10259u, 10260u, 12419u})
    {
      const BatchWhere at = batch_locate(lay.descs.data(), 3, lay.uniform, lay.pp, lay.first8, w0);
      CHECK(at.p == (w0 < 8100 ? 0u : (w0 < 10260 ? 1u : 2u)) && at.tile < lay.descs[at.p].tiles && at.row < lay.descs[at.p].rows);
    }
  }
  { // paired rows (k_u8_batch, k_q32_batch): the 8K 4:2:0 frame takes 12,150 tiles instead of 12,420; odd row counts leave a single last row;
    // planes that do not end in half a tile, single-row planes and huge pitches stay unpaired; equal paired planes are still `uniform`
    std::vector<mdct_plane_u8> planes(3);
    const size_t w[3] = {7680, 3840, 3840}, h[3] = {4320, 2160, 2160};
    for (int i = 0; i < 3; i++)
      planes[i] = mdct_plane_u8{reinterpret_cast<const uint8_t *>((uintptr_t)0x1000 * (i + 1)), reinterpret_cast<uint8_t *>((uintptr_t)0x100000 * (i + 1)), w[i] + 16, w[i], w[i], h[i], nullptr};
    const int ids[] = {0, 1, 1};
    const unsigned char has[] = {1, 1, 1};
    BatchLayout lay;
    batch_layout(planes.data(), ids, has, 0, 3, 3584, 512, lay, true);
    CHECK(lay.consumed == 3 && lay.total == 15u * 540 + 2 * 15u * 135 && lay.total == 12150u && !lay.uniform);
    CHECK(!(lay.descs[0].has_lut & kDescPaired) && (lay.descs[1].has_lut & kDescPaired) && (lay.descs[1].has_lut & kDescLut) && lay.descs[1].tiles == 15);
    CHECK(walk(lay, planes) == 12150u);
    std::mt19937 rng(11);
    for (int n : {1, 2, 5, 8, 9, 20})
    {
      std::vector<mdct_plane_u8> ps;
      std::vector<int> id;
      std::vector<unsigned char> hs;
      int expect_paired = 0;
      for (int i = 0; i < n; i++)
      {
        const size_t bpr = (i % 3 == 0) ? 32 + 64 * (rng() % 6) : 1 + rng() % 300, rows = 1 + rng() % 9;
        const size_t pitch = (i % 7 == 6) ? ((size_t)1 << 26) : bpr * 8 + 8 * (rng() % 3);
        ps.push_back(mdct_plane_u8{reinterpret_cast<const uint8_t *>((uintptr_t)0x10000 * (i + 1)), reinterpret_cast<uint8_t *>((uintptr_t)0x1000000 * (i + 1)), pitch, pitch, bpr * 8, rows * 8, nullptr});
        id.push_back(i % 2);
        hs.push_back(1);
        expect_paired += (bpr % 64 == 32 && rows >= 2 && pitch < ((size_t)1 << 26)) ? 1 : 0;
      }
      BatchLayout l2;
      batch_layout(ps.data(), id.data(), hs.data(), 0, n, 0, 512, l2, true);
      CHECK(l2.consumed == n);
      int got_paired = 0;
      for (auto &d : l2.descs)
        got_paired += (d.has_lut & kDescPaired) ? 1 : 0;
      CHECK(got_paired == expect_paired);
      walk(l2, ps);
      BatchLayout l3; // the same planes for a kernel that does not pair: one tile grid per row, as before
      batch_layout(ps.data(), id.data(), hs.data(), 0, n, 0, 512, l3);
      for (auto &d : l3.descs)
        CHECK(!(d.has_lut & kDescPaired) && d.tiles == (d.bpr + 63) / 64);
      walk(l3, ps);
      CHECK(l3.total >= l2.total);
    }
    std::vector<mdct_plane_u8> same(6, mdct_plane_u8{reinterpret_cast<const uint8_t *>((uintptr_t)0x10000), reinterpret_cast<uint8_t *>((uintptr_t)0x1000000), 3840, 3840, 3840, 8 * 7, nullptr});
    const int ids6[] = {0, 0, 0, 0, 0, 0};
    const unsigned char has6[] = {1, 1, 1, 1, 1, 1};
    batch_layout(same.data(), ids6, has6, 0, 6, 0, 512, lay, true);
    CHECK(lay.uniform && lay.per_plane == 3u * 15 + 8 && lay.total == 6u * 53);
    walk(lay, same);
  }
  puts("batch plan ok");
  return 0;
}
