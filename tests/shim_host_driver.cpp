// shim_host_driver.cpp -- TEST INFRASTRUCTURE: the host-only logic of the drop-in shim (csrc/shim_host.h: ref_range,
// level_from_flags, CopyPool, StripPipeline) on a back end made of worker-thread "streams" and sequence-number "events" over plain heap memory,
// built by tests/test_shim_host.py with -fsanitize=thread and with -fsanitize=address,undefined.  Every buffer is an
// exact-size heap allocation that is freed the moment the pipeline returns, so a job that outlives its call, a copy
// past a strip or a race between the helpers and the caller is a sanitizer report.
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <memory>
#include <map>
#include <vector>

#include "shim_host.h"

using namespace mdct_host;

#define CHECK(c)                                                     \
  do                                                                 \
  {                                                                  \
    if (!(c))                                                        \
    {                                                                \
      fprintf(stderr, "%s:%d: CHECK failed: %s\n", __FILE__, __LINE__, #c); \
      exit(1);                                                       \
    }                                                                \
  } while (0)

// a "stream": one worker thread running queued closures in order
struct FakeStream
{
  std::mutex m;
  std::condition_variable cv, idle;
  std::deque<std::function<void()>> q;
  bool stop = false, busy = false;
  std::thread th;
  FakeStream() : th([this] { loop(); }) {}
  ~FakeStream()
  {
    {
      std::lock_guard<std::mutex> lk(m);
      stop = true;
    }
    cv.notify_all();
    th.join();
  }
  void loop()
  {
    for (;;)
    {
      std::function<void()> f;
      {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return stop || !q.empty(); });
        if (q.empty())
          return;
        f = std::move(q.front());
        q.pop_front();
        busy = true;
      }
      f();
      {
        std::lock_guard<std::mutex> lk(m);
        busy = false;
      }
      idle.notify_all();
    }
  }
  void push(std::function<void()> f)
  {
    {
      std::lock_guard<std::mutex> lk(m);
      q.push_back(std::move(f));
    }
    cv.notify_one();
  }
  void sync()
  {
    std::unique_lock<std::mutex> lk(m);
    idle.wait(lk, [&] { return q.empty() && !busy; });
  }
};

// an "event": a point in a stream's order.  record = bump the sequence now and mark it reached when the stream gets there; a waiter
// (host thread or another stream) takes the sequence number at the time of ITS call, like hipEventSynchronize / hipStreamWaitEvent
struct FakeEvent
{
  std::mutex m;
  std::condition_variable cv;
  uint64_t recorded = 0, reached = 0;
  void reach(uint64_t seq)
  {
    {
      std::lock_guard<std::mutex> lk(m);
      reached = seq > reached ? seq : reached;
    }
    cv.notify_all();
  }
  void wait_for(uint64_t seq)
  {
    std::unique_lock<std::mutex> lk(m);
    cv.wait(lk, [&] { return reached >= seq; });
  }
  uint64_t target()
  {
    std::lock_guard<std::mutex> lk(m);
    return recorded;
  }
};

static uint8_t kernel_byte(uint8_t in, size_t i) { return (uint8_t)(in * 31u + (uint8_t)(i * 7u) + (uint8_t)(i >> 11)); }

// The planes of a call as the pipeline sees them (shim_host.h: Pieces), and the stand-in kernel: block row r gathers its in.count pieces of
// in.row bytes, maps every byte with its position, and scatters the result over its out.count pieces of out.row bytes.  keep / period:
// of every `period` output bytes the kernel writes the first `keep` and rubbish into the rest of the DEVICE plane (the SSE encq tier
// writes half of every block pair, simd_dct.cpp:1662-1676); tail: it also writes `tail` bytes behind its last row (the spill, :1676).
struct Geometry
{
  Pieces in, out;
  size_t rows;               // block rows of the planes
  size_t keep = 0, period = 0, tail = 0;
  size_t row_bytes() const { return in.count * in.row; }
  size_t in_bytes() const { return (in.count - 1) * in.stride + rows * in.row; }
  size_t out_bytes() const { return (out.count - 1) * out.stride + rows * out.row + tail; }
};
static Geometry strips(size_t strip, size_t rows) { return {{1, 0, strip}, {1, 0, strip}, rows}; }
static Geometry half_pairs(size_t strip, size_t rows) { return {{1, 0, strip}, {1, 0, strip}, rows, 64, 128, 64}; }
// 2 stacked input images, 16 output planes (the stereo layout has 64; the pipeline only sees "several")
static Geometry stereo_like(size_t strip, size_t rows) { return {{2, rows * strip + 192, strip}, {16, rows * strip / 8 + 64, strip / 8}, rows}; }

static uint8_t spill_byte(size_t r1, size_t k) { return (uint8_t)(0x40 + r1 * 3 + k); }
static void kernel_rows(const Geometry &g, const uint8_t *in, uint8_t *out, size_t r0, size_t r1)
{
  for (size_t r = r0; r < r1; r++)
    for (size_t k = 0; k < g.row_bytes(); k++)
    {
      const uint8_t v = kernel_byte(in[(k / g.in.row) * g.in.stride + r * g.in.row + k % g.in.row], r * g.row_bytes() + k);
      const size_t o = k % g.out.row;
      out[(k / g.out.row) * g.out.stride + r * g.out.row + o] = (g.period && o % g.period >= g.keep) ? (uint8_t)0xEE : v;
    }
  for (size_t k = 0; k < g.tail && r1 > r0; k++)
    out[r1 * g.out.row + k] = spill_byte(r1, k); // (overwritten by the next chunk's kernel, except behind the call's last row)
}

struct FakeDev
{
  typedef FakeStream *stream_t;
  const uint8_t *d_in = nullptr;
  uint8_t *d_out = nullptr;
  Geometry geo;
  // failure injection: the n-th call of each kind fails (counted from 0; -1 = never)
  std::atomic<int> h2d_calls{0}, d2h_calls{0}, launch_calls{0}, wait_calls{0};
  int fail_h2d = -1, fail_d2h = -1, fail_launch = -1, fail_wait = -1;
  std::atomic<int> bound_threads{0};

  typedef FakeEvent event_t;
  void bind_thread() { bound_threads++; }
  bool stream_wait(stream_t s)
  {
    s->sync();
    return wait_calls++ != fail_wait;
  }
  bool event_record(FakeEvent &e, stream_t s)
  {
    uint64_t seq;
    {
      std::lock_guard<std::mutex> lk(e.m);
      seq = ++e.recorded;
    }
    FakeEvent *pe = &e;
    s->push([=] { pe->reach(seq); });
    return true;
  }
  bool stream_wait_event(stream_t s, FakeEvent &e)
  {
    const uint64_t seq = e.target();
    FakeEvent *pe = &e;
    s->push([=] { pe->wait_for(seq); });
    return true;
  }
  bool event_wait(FakeEvent &e)
  {
    e.wait_for(e.target());
    return wait_calls++ != fail_wait;
  }
  bool h2d_async(uint8_t *dev, const uint8_t *host, size_t n, stream_t s) { return h2d_2d_async(dev, n, host, n, n, 1, s); }
  bool d2h_async(uint8_t *host, const uint8_t *dev, size_t n, stream_t s) { return d2h_2d_async(host, n, dev, n, n, 1, s); }
  bool h2d_2d_async(uint8_t *dev, size_t dpitch, const uint8_t *host, size_t spitch, size_t width, size_t height, stream_t s)
  {
    if (h2d_calls++ == fail_h2d)
      return false;
    s->push([=] { copy_pieces(dev, dpitch, host, spitch, width, height, 0, 0); });
    return true;
  }
  bool d2h_2d_async(uint8_t *host, size_t dpitch, const uint8_t *dev, size_t spitch, size_t width, size_t height, stream_t s)
  {
    if (d2h_calls++ == fail_d2h)
      return false;
    s->push([=] { copy_pieces(host, dpitch, dev, spitch, width, height, 0, 0); });
    return true;
  }
  int launch(size_t r0, size_t r1, stream_t s)
  {
    if (launch_calls++ == fail_launch)
      return 2; // "not supported", like a failed kernel launch
    const uint8_t *in = d_in;
    uint8_t *out = d_out;
    const Geometry g = geo;
    s->push([=] { kernel_rows(g, in, out, r0, r1); });
    return 0;
  }
};

// one calling thread's staging, as thread_local Staging of shim.hip: everything exact-size on the heap
struct Rig
{
  Geometry geo;
  size_t rpc;
  std::unique_ptr<uint8_t[]> d_in, d_out, pin_in[kPipeSlots], pin_out[kPipeSlots];
  uint8_t *pin_in_p[kPipeSlots], *pin_out_p[kPipeSlots];
  FakeStream streams[3];
  FakeEvent e_in[kPipeSlots], e_k[kPipeSlots], e_out[kPipeSlots];
  FakeDev dev;
  CopyPool<FakeDev> pool_in, pool_out;
  std::atomic<int> in_latch[kPipeSlots] = {}, out_latch[kPipeSlots] = {};
  Rig(const Geometry &g, size_t rpc_) : geo(g), rpc(rpc_)
  {
    d_in.reset(new uint8_t[g.in_bytes()]);
    d_out.reset(new uint8_t[g.out_bytes()]);
    for (int i = 0; i < kPipeSlots; i++)
    {
      pin_in[i].reset(new uint8_t[g.row_bytes() * rpc]);
      pin_out[i].reset(new uint8_t[g.row_bytes() * rpc]);
      pin_in_p[i] = pin_in[i].get();
      pin_out_p[i] = pin_out[i].get();
    }
    dev.d_in = d_in.get();
    dev.d_out = d_out.get();
    dev.geo = g;
  }
  Rig(size_t strip, size_t rows, size_t rpc_) : Rig(strips(strip, rows), rpc_) {}
  int run(const uint8_t *from, uint8_t *to, size_t b0, size_t b1, bool helpers, bool pinned_in, bool pinned_out)
  {
    StripPipeline<FakeDev> pl{&dev, &pool_in, &pool_out, from, to, d_in.get(), d_out.get(), pin_in_p, pin_out_p, &streams[0], &streams[1], &streams[2],
                              e_in, e_k, e_out, in_latch, out_latch, geo.in, geo.out, rpc, pinned_in, pinned_out, helpers};
    pl.out_keep = geo.keep;
    pl.out_period = geo.period;
    pl.out_tail = b1 > b0 && b1 < geo.rows + (geo.tail ? 1 : 0) ? geo.tail : 0; // (the planes of these tests hold the last row's spill too)
    return pl.run(b0, b1);
  }
  bool quiescent()
  {
    std::lock_guard<std::mutex> a(pool_in.m), b(pool_out.m);
    bool q = pool_in.q.empty() && pool_out.q.empty();
    for (int i = 0; i < kPipeSlots; i++)
      q = q && in_latch[i] == 0 && out_latch[i] == 0;
    return q;
  }
};

static std::vector<uint8_t> make_plane(size_t n, unsigned seed)
{
  std::vector<uint8_t> v(n);
  unsigned x = seed * 2654435761u + 12345u;
  for (size_t i = 0; i < n; i++)
  {
    x = x * 1664525u + 1013904223u;
    v[i] = (uint8_t)(x >> 24);
  }
  return v;
}

// what a call on block rows [b0, b1) must leave in the caller's output plane, byte for byte: the kernel's bytes where it writes, the
// caller's canary everywhere else (other rows, the gaps between pieces, the unwritten half of every pair)
static void check_geo(const Geometry &g, const std::vector<uint8_t> &in, const uint8_t *out, size_t b0, size_t b1, uint8_t canary)
{
  std::vector<uint8_t> dev_out(g.out_bytes(), 0), want(g.out_bytes(), canary);
  kernel_rows(g, in.data(), dev_out.data(), b0, b1);
  for (size_t p = 0; p < g.out.count; p++)
    for (size_t o = b0 * g.out.row; o < b1 * g.out.row; o++)
      if (!g.period || (o % g.out.row) % g.period < g.keep)
        want[p * g.out.stride + o] = dev_out[p * g.out.stride + o];
  for (size_t k = 0; k < g.tail && b1 > b0; k++)
    want[b1 * g.out.row + k] = spill_byte(b1, k);
  for (size_t i = 0; i < g.out_bytes(); i++)
    if (out[i] != want[i])
    {
      fprintf(stderr, "byte %zu: got %u want %u (rows [%zu, %zu))\n", i, out[i], want[i], b0, b1);
      exit(1);
    }
}
static void check_rows(const std::vector<uint8_t> &in, const uint8_t *out, size_t strip, size_t rows, size_t b0, size_t b1, uint8_t canary)
{
  check_geo(strips(strip, rows), in, out, b0, b1, canary);
}

static void test_single_caller()
{
  const size_t strip = 1024;
  for (size_t rows : {1, 2, 3, 7, 16, 29})
    for (size_t rpc : {1, 2, 3})
      for (int helpers = 0; helpers < 2; helpers++)
        for (int pins = 0; pins < 4; pins++)
        {
          const std::vector<uint8_t> in = make_plane(strip * rows, (unsigned)(rows * 16 + rpc));
          std::unique_ptr<uint8_t[]> out(new uint8_t[strip * rows]);
          for (size_t b0 = 0; b0 < rows; b0 += (rows > 4 ? 3 : 1))
          {
            memset(out.get(), 0xC3, strip * rows);
            Rig rig(strip, rows, rpc);
            CHECK(rig.run(in.data(), out.get(), b0, rows, helpers, pins & 1, pins & 2) == PIPELINE_OK);
            CHECK(rig.quiescent());
            check_rows(in, out.get(), strip, rows, b0, rows, 0xC3);
          }
        }
  puts("single caller: ok");
}

// the layouts whose chunks are not one strip: several pieces per direction (stereo: 2-D copies, helpers sharing by piece or by byte range),
// and the half-written pairs with their trailing spill (only the kernel's bytes reach the caller, never the rubbish between them)
static void test_pieces_and_half_pairs()
{
  for (int kind = 0; kind < 2; kind++)
    for (size_t rows : {1, 2, 5, 13, 22})
      for (size_t rpc : {1, 2, 3})
        for (int helpers = 0; helpers < 2; helpers++)
          for (int pins = 0; pins < (kind == 0 ? 4 : 2); pins++) // (half pairs: the output always goes through the bounce buffers)
          {
            const Geometry g = kind == 0 ? stereo_like(1024, rows) : half_pairs(1024, rows);
            const std::vector<uint8_t> in = make_plane(g.in_bytes(), (unsigned)(rows * 8 + rpc + kind));
            std::unique_ptr<uint8_t[]> out(new uint8_t[g.out_bytes()]);
            for (size_t b0 = 0; b0 < rows; b0 += (rows > 4 ? 4 : 1))
              for (size_t b1 : {rows, (b0 + rows + 1) / 2})
              {
                if (b1 <= b0)
                  continue;
                memset(out.get(), 0x7E, g.out_bytes());
                Rig rig(g, rpc);
                CHECK(rig.run(in.data(), out.get(), b0, b1, helpers, pins & 1, pins & 2) == PIPELINE_OK);
                CHECK(rig.quiescent());
                check_geo(g, in, out.get(), b0, b1, 0x7E);
              }
          }
  puts("several pieces per chunk, half-written pairs with their spill: ok");
}

// the reference's intended multi-core use: concurrent calls on disjoint row ranges of the SAME planes
static void test_four_callers()
{
  const size_t strip = 4096, rows = 64, rpc = 3, T = 4;
  const std::vector<uint8_t> in = make_plane(strip * rows, 77);
  std::unique_ptr<uint8_t[]> out(new uint8_t[strip * rows]);
  memset(out.get(), 0x5A, strip * rows);
  std::vector<std::thread> th;
  std::atomic<int> bad{0};
  for (size_t t = 0; t < T; t++)
    th.emplace_back([&, t] {
      Rig rig(strip, rows, rpc); // per-thread staging, like thread_local Staging
      for (int rep = 0; rep < 3; rep++)
        if (rig.run(in.data(), out.get(), rows * t / T, rows * (t + 1) / T, true, false, false) != PIPELINE_OK)
          bad++;
      if (!rig.quiescent())
        bad++;
    }); // the Rig dies here: helper threads joined, buffers freed
  for (auto &x : th)
    x.join();
  CHECK(bad == 0);
  check_rows(in, out.get(), strip, rows, 0, rows, 0x5A);
  puts("four callers on disjoint ranges: ok");
}

// early-error paths: whatever fails, on return nothing is queued, no latch is up, both streams are idle -- the
// buffers are freed right after the call, so a straggler would be a use-after-free / race report
static void test_failures()
{
  const size_t strip = 2048, rows = 13, rpc = 2; // 7 chunks: the slots are reused
  for (int geo = 0; geo < 3; geo++)
  for (int helpers = 0; helpers < 2; helpers++)
    for (int kind = 0; kind < 4; kind++)
      for (int at = 0; at < 8; at++)
      {
        const Geometry g = geo == 0 ? strips(strip, rows) : (geo == 1 ? stereo_like(strip, rows) : half_pairs(strip, rows));
        const std::vector<uint8_t> in = make_plane(g.in_bytes(), 5);
        std::unique_ptr<uint8_t[]> out(new uint8_t[g.out_bytes()]);
        memset(out.get(), 0x11, g.out_bytes());
        int r;
        {
          Rig rig(g, rpc);
          if (kind == 0) rig.dev.fail_h2d = at;
          if (kind == 1) rig.dev.fail_d2h = at;
          if (kind == 2) rig.dev.fail_launch = at;
          if (kind == 3) rig.dev.fail_wait = at;
          r = rig.run(in.data(), out.get(), 0, rows, helpers, false, false);
          CHECK(rig.quiescent());
          for (FakeStream &st : rig.streams)
          {
            std::lock_guard<std::mutex> a(st.m);
            CHECK(st.q.empty() && !st.busy);
          }
          if (kind == 2)
            CHECK(r == (at < 7 ? 2 : PIPELINE_OK)); // 7 chunks = 7 launches; (at == 7 is there for the half pairs' eighth copy out: the spill)
          else
            CHECK(r == PIPELINE_FAILED || r == PIPELINE_OK); // a failing wait during abandon()'s own drain does not change a success
        } // rig and its buffers are gone
        out.reset();
      }
  puts("early-error paths: ok");
}

// a thread that exits with jobs still queued: shutdown() serves them all before the helpers leave
static void test_exit_with_jobs_queued()
{
  for (int rep = 0; rep < 20; rep++)
  {
    FakeDev dev;
    FakeStream s;
    std::vector<uint8_t> src = make_plane(1 << 16, (unsigned)rep), dst(1 << 16, 0);
    std::atomic<int> latch{64};
    {
      CopyPool<FakeDev> pool;
      CHECK(pool.start(&dev));
      for (int i = 0; i < 64; i++)
        pool.push({(i & 1) != 0, &s, nullptr, dst.data() + i * 1024, src.data() + i * 1024, 1024, &latch});
    } // ~CopyPool with most of the 64 jobs still queued
    CHECK(latch == 0 && dst == src);
    CHECK(dev.bound_threads == CopyPool<FakeDev>::kThreads);
  }
  puts("exit with jobs queued: ok");
}

// ref_range against the reference's own loop (simd_dct.cpp:2243-2261 with y*2, :375-387 without)
static void test_ref_range()
{
  for (size_t sizeY = 0; sizeY <= 160; sizeY += 8)
    for (size_t startY = 0; startY <= sizeY + 24; startY++)
      for (size_t endY = 0; endY <= sizeY + 24; endY++)
        for (size_t step : {(size_t)16, (size_t)8})
        {
          std::vector<size_t> rows;
          for (size_t y = 0; y < sizeY / 2; y += 8)
          {
            const size_t yy = step == 16 ? y * 2 : y;
            if (yy < startY)
              continue;
            else if (yy > endY)
              break;
            rows.push_back(y / 8);
          }
          size_t b0, b1;
          ref_range(sizeY, startY, endY, step, &b0, &b1);
          CHECK(b1 - b0 == rows.size());
          if (!rows.empty())
            CHECK(rows.front() == b0 && rows.back() + 1 == b1);
        }
  puts("ref_range == the reference's loop: ok");
}

// level_from_flags against the three dispatchers' if-chains (simd_dct.cpp:78-85, :100-105, :120-127)
static void test_levels()
{
  for (int m = 0; m < 32; m++)
  {
    const bool sse2 = m & 1, ssse3 = m & 2, sse41 = m & 4, avx2 = m & 8, avx512vl = m & 16;
    const int lv = level_from_flags(-1, &sse2, &ssse3, &sse41, &avx2, &avx512vl);
    // q32 (:120-127): AVX-512VL -> AVX2 -> [SSE4.1 variant, not reproduced] -> not supported
    CHECK((lv >= LEVEL_AVX2) == (avx512vl || avx2));
    if (!(avx512vl || avx2))
    {
      // stereo (:78-85): sse41&&sse2 -> ssse3&&sse2 -> sse2 -> scalar: SSE bytes iff any of them
      const bool stereo_sse = (sse41 && sse2) || (ssse3 && sse2) || sse2;
      CHECK((lv >= LEVEL_SSE2) == stereo_sse);
      // encq (:100-105): sse41&&sse2 -> ssse3&&sse2 -> scalar
      const bool encq_sse = (sse41 && sse2) || (ssse3 && sse2);
      CHECK((lv >= LEVEL_SSSE3) == encq_sse);
    }
  }
  CHECK(level_from_flags(-1, nullptr, nullptr, nullptr, nullptr, nullptr) == LEVEL_AVX2);
  const bool f = false;
  CHECK(level_from_flags(LEVEL_SSE2, &f, &f, &f, &f, &f) == LEVEL_SSE2);
  CHECK(level_from_flags(-1, &f, &f, &f, &f, &f) == LEVEL_NONE); // all-false until _DetectCPUFeatures()
  puts("tier choice == the dispatchers: ok");
}


// ---- AutoPin (MDCT_SHIM_AUTOPIN): a back end that models the runtime's rule -- ranges may not overlap, only a registered base can be
// released -- and fails on request
struct PinDev
{
  std::mutex m;
  std::map<uintptr_t, size_t> live;
  int fail_next = 0;
  int registers = 0, unregisters = 0;
  bool host_register(void *p, size_t n)
  {
    std::lock_guard<std::mutex> lk(m);
    if (fail_next > 0)
    {
      fail_next--;
      return false;
    }
    const uintptr_t b = (uintptr_t)p;
    for (auto &kv : live)
      CHECK(!(kv.first < b + n && b < kv.first + kv.second)); // AutoPin must never ask for an overlapping range
    live[b] = n;
    registers++;
    return true;
  }
  bool host_unregister(void *p)
  {
    std::lock_guard<std::mutex> lk(m);
    CHECK(live.erase((uintptr_t)p) == 1);
    unregisters++;
    return true;
  }
};

static void test_autopin()
{
  typedef AutoPin<PinDev> AP;
  std::vector<uint8_t> a(1 << 16), b(1 << 16);
  { // third sighting registers; later calls ride it; a longer reach re-registers once nobody relies on the shorter one
    PinDev dev;
    AP ap;
    CHECK(ap.enter(dev, a.data(), 4096) == -1 && ap.enter(dev, a.data(), 4096) == -1 && dev.registers == 0);
    const int h = ap.enter(dev, a.data(), 4096);
    CHECK(h >= 0 && dev.registers == 1 && dev.live.at((uintptr_t)a.data()) == 4096);
    const int h2 = ap.enter(dev, a.data(), 1000); // shorter: covered
    CHECK(h2 == h && dev.registers == 1);
    CHECK(ap.enter(dev, a.data(), 8192) == -1 && dev.registers == 1); // longer while two calls rely on the registration: left alone
    ap.leave(h);
    ap.leave(h2);
    const int h3 = ap.enter(dev, a.data(), 8192);
    CHECK(h3 >= 0 && dev.registers == 2 && dev.unregisters == 1 && dev.live.at((uintptr_t)a.data()) == 8192);
    ap.leave(h3);
    // an overlapping range from another base is never registered
    for (int i = 0; i < 5; i++)
      CHECK(ap.enter(dev, a.data() + 4096, 8192) == -1);
    CHECK(dev.registers == 2);
    // null / empty
    CHECK(ap.enter(dev, nullptr, 10) == -1 && ap.enter(dev, a.data(), 0) == -1);
    CHECK(ap.release_all(dev) == 0 && dev.live.empty() && dev.unregisters == 2);
    CHECK(ap.enter(dev, a.data(), 64) == -1); // forgotten: three sightings again
  }
  { // a failed registration is not retried; release_all keeps what a running call relies on
    PinDev dev;
    AP ap;
    dev.fail_next = 1;
    for (int i = 0; i < 6; i++)
      CHECK(ap.enter(dev, a.data(), 4096) == -1);
    CHECK(dev.registers == 0);
    int h = -1;
    for (int i = 0; i < 3; i++)
      h = ap.enter(dev, b.data(), 4096);
    CHECK(h >= 0);
    CHECK(ap.release_all(dev) == 1 && dev.live.size() == 1); // in use
    ap.leave(h);
    CHECK(ap.release_all(dev) == 0 && dev.live.empty());
  }
  { // more ranges than entries: the least recently used registration goes, never one in use
    PinDev dev;
    AP ap;
    std::vector<std::vector<uint8_t>> bufs(AP::kRanges + 4, std::vector<uint8_t>(256));
    int held = -1;
    for (int i = 0; i < 3; i++)
      held = ap.enter(dev, bufs[0].data(), 256);
    CHECK(held >= 0); // stays in use throughout
    for (size_t k = 1; k < bufs.size(); k++)
      for (int i = 0; i < 3; i++)
        ap.leave(ap.enter(dev, bufs[k].data(), 256));
    CHECK(dev.live.count((uintptr_t)bufs[0].data()) == 1 && (int)dev.live.size() <= AP::kRanges && dev.unregisters >= 4);
    ap.leave(held);
    CHECK(ap.release_all(dev) == 0 && dev.live.empty() && dev.registers == dev.unregisters);
  }
  { // 8 threads on disjoint row ranges of the same two planes (the reference's startY/endY use), a ninth releasing now and then
    PinDev dev;
    AP ap;
    std::atomic<bool> stop{false};
    std::vector<std::thread> ts;
    for (int t = 0; t < 8; t++)
      ts.emplace_back([&, t] {
        for (int i = 0; i < 400; i++)
        {
          const size_t reach = 4096 * (size_t)(t + 1);
          const int hi = ap.enter(dev, a.data(), reach), ho = ap.enter(dev, b.data(), reach);
          if (hi >= 0)
          {
            std::lock_guard<std::mutex> lk(dev.m);
            auto it = dev.live.find((uintptr_t)a.data());
            CHECK(it != dev.live.end() && it->second >= reach); // what the call relies on is really there
          }
          ap.leave(hi);
          ap.leave(ho);
        }
      });
    std::thread rel([&] {
      while (!stop)
      {
        (void)ap.release_all(dev);
        std::this_thread::yield();
      }
    });
    for (auto &t : ts)
      t.join();
    stop = true;
    rel.join();
    CHECK(ap.release_all(dev) == 0 && dev.live.empty() && dev.registers == dev.unregisters);
  }
}

int main()
{
  test_autopin();
  test_ref_range();
  test_levels();
  test_single_caller();
  test_pieces_and_half_pairs();
  test_four_callers();
  test_failures();
  test_exit_with_jobs_queued();
  puts("shim host ok");
  return 0;
}
