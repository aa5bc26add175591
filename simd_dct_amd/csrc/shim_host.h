// shim_host.h -- the HOST-ONLY logic of the drop-in shim (csrc/shim.hip), free of any HIP type so that it also builds
// with plain g++: the reference's row-range arithmetic, the tier choice from the reference's CPU-flag globals, the
// helper-thread copy pools and the chunked strip pipeline (one stream per stage, four slots) that serves host-pointer calls.  shim.hip
// instantiates the templates with the HIP runtime as back end; tests/shim_host_driver.cpp instantiates them with
// worker-thread "streams" over plain memory and runs them under -fsanitize=thread and -fsanitize=address,undefined
// (the GPU pool cannot run sanitizers).
//
// Back end contract (`Dev`):
//   typedef ... stream_t;                                        copyable handle, value-initialisable to "none"
//   typedef ... event_t;                                         a point in a stream's order; owned by the caller of the pipeline
//   void bind_thread();                                          called once by every helper thread before it works
//   bool stream_wait(stream_t);                                  block until everything queued on the stream is done
//   bool event_record(event_t &, stream_t);                      (re-)record the event behind everything queued on the stream so far
//   bool stream_wait_event(stream_t, event_t &);                 later work on the stream starts after the event's last record (no host block)
//   bool event_wait(event_t &);                                  block the calling host thread until the event's last record is reached
//   bool h2d_async(uint8_t *dev, const uint8_t *host, size_t n, stream_t);
//   bool d2h_async(uint8_t *host, const uint8_t *dev, size_t n, stream_t);
//   int  launch(size_t row0, size_t row1, stream_t);             block rows [row0, row1) dev-in -> dev-out; 0 = ok
//   bool host_register(void *, size_t); bool host_unregister(void *);   (AutoPin only) page-lock / release a range of the caller's memory
#ifndef MDCT_SHIM_HOST_H
#define MDCT_SHIM_HOST_H

#include <atomic>
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

namespace mdct_host
{

inline size_t ceil_div(size_t a, size_t b) { return (a + b - 1) / b; }

// Block rows the reference's loop `for (y = 0; y < sizeY/2; y += 8)` processes (simd_dct.cpp:2243-2261):
// step = 16 for the `y*2` tiers (processed iff startY <= 2y <= endY), 8 for the scalar encq tier (:375-387).
inline void ref_range(size_t sizeY, size_t startY, size_t endY, size_t step, size_t *b0, size_t *b1)
{
  const size_t rows = ceil_div(sizeY / 2, 8);
  *b0 = ceil_div(startY, step);
  const size_t last = endY / step + 1;
  *b1 = last < rows ? last : rows;
  if (*b0 > *b1)
    *b0 = *b1;
}

// tier levels, == MDCT_SIMD_* of include/simd_dct_shim.h (static_asserted in shim.hip)
enum { LEVEL_NONE = 0, LEVEL_SSE2 = 1, LEVEL_SSSE3 = 2, LEVEL_SSE41 = 3, LEVEL_AVX2 = 4 };

// What the reference's dispatchers would pick from its CPU-flag globals (simd_dct.cpp:78-85, :100-105, :120-127);
// null pointers = the program does not link simd_platform.c -> the AVX2 tier.
inline int level_from_flags(int set_level, const bool *sse2, const bool *ssse3, const bool *sse41, const bool *avx2, const bool *avx512vl)
{
  if (set_level >= 0)
    return set_level;
  if (sse2 && ssse3 && sse41 && avx2 && avx512vl)
  {
    if (*avx512vl || *avx2)
      return LEVEL_AVX2;
    if (*sse41 && *sse2)
      return LEVEL_SSE41;
    if (*ssse3 && *sse2)
      return LEVEL_SSSE3;
    if (*sse2)
      return LEVEL_SSE2;
    return LEVEL_NONE;
  }
  return LEVEL_AVX2;
}

// Opt-in auto-pinning (MDCT_SHIM_AUTOPIN=1; shim.hip).  A caller like the reference's harness hands the SAME two pageable buffers to every
// call (main.cpp:510-523 reuses them for all --runs): the third time a host range is seen it is page-locked in place (host_register), and
// from then on the pipeline DMAs straight from / to it instead of going through memcpy and the bounce buffers -- what mdct_shim_pin() does
// by hand.  One registry per process (calls on disjoint row ranges of one plane come from many threads): at most kRanges ranges, least
// recently used released first, never one that a running call relies on (`users`); everything is released by release_all() --
// mdct_shim_release(), a calling thread's exit, process exit.  The caller's side of the bargain (why this is opt-in): a buffer it has passed
// three times stays allocated until then -- a range that is freed and re-allocated at the same address while registered would be DMA'd
// through its old pages.
template <class Dev>
struct AutoPin
{
  static constexpr int kRanges = 16;
  static constexpr unsigned kThreshold = 3;
  struct Range
  {
    uintptr_t base = 0;
    size_t len = 0;      // the longest extent a call has touched from `base`
    size_t pinned_len = 0; // registered bytes (0: not registered)
    unsigned seen = 0;
    int users = 0;       // calls between enter() and leave()
    bool used = false, dead = false; // dead: registering failed once, never tried again
    uint64_t tick = 0;
  };
  std::mutex mu;
  Range r[kRanges];
  uint64_t tick = 0;
  uint64_t registered = 0, unregistered = 0; // counters (tests, diagnostics)

  // A call is about to touch [p, p + len).  Returns a handle >= 0 when the range is page-locked by this registry and will stay so until
  // leave(handle); -1 otherwise (not seen often enough yet, could not be registered, or overlaps another registered range).
  int enter(Dev &dev, const void *p, size_t len)
  {
    if (p == nullptr || len == 0)
      return -1;
    const uintptr_t base = (uintptr_t)p;
    std::lock_guard<std::mutex> lk(mu);
    int k = -1;
    for (int i = 0; i < kRanges && k < 0; i++)
      if (r[i].used && r[i].base == base)
        k = i;
    if (k < 0)
    { // a free entry, or the least recently used one nobody relies on
      int v = -1;
      for (int i = 0; i < kRanges; i++)
        if (!r[i].used)
        {
          v = i;
          break;
        }
        else if (r[i].users == 0 && (v < 0 || r[i].tick < r[v].tick))
          v = i;
      if (v < 0)
        return -1;
      if (r[v].used && r[v].pinned_len)
        drop(dev, r[v]);
      r[v] = Range();
      r[v].used = true;
      r[v].base = base;
      k = v;
    }
    Range &e = r[k];
    e.tick = ++tick;
    e.seen++;
    if (len > e.len)
      e.len = len;
    if (e.pinned_len >= len)
    {
      e.users++;
      return k;
    }
    if (e.dead || e.seen < kThreshold)
      return -1;
    if (e.pinned_len)
    { // registered, but this call reaches further: once nobody relies on the shorter registration, replace it
      if (e.users)
        return -1;
      drop(dev, e);
    }
    for (int i = 0; i < kRanges; i++) // (host_register refuses overlapping ranges; so do we, before asking)
      if (i != k && r[i].used && r[i].pinned_len && r[i].base < base + e.len && base < r[i].base + r[i].pinned_len)
        return -1;
    if (!dev.host_register((void *)base, e.len))
    {
      e.dead = true;
      return -1;
    }
    registered++;
    e.pinned_len = e.len;
    e.users++;
    return k;
  }
  void leave(int handle)
  {
    if (handle < 0)
      return;
    std::lock_guard<std::mutex> lk(mu);
    r[handle].users--;
  }
  // releases every registration no running call relies on and forgets the sightings; returns the number still held
  int release_all(Dev &dev)
  {
    std::lock_guard<std::mutex> lk(mu);
    int held = 0;
    for (int i = 0; i < kRanges; i++)
    {
      if (!r[i].used)
        continue;
      if (r[i].users)
      {
        held++;
        continue;
      }
      if (r[i].pinned_len)
        drop(dev, r[i]);
      r[i] = Range();
    }
    return held;
  }

private:
  void drop(Dev &dev, Range &e)
  {
    (void)dev.host_unregister((void *)e.base);
    unregistered++;
    e.pinned_len = 0;
  }
};

// `count` pieces of `len` bytes, `dpitch` / `spitch` apart; period != 0: of every `period` bytes of a piece only the first `keep` are copied
// (the SSE encq tier writes the first 64 bytes of every 128-byte block pair, simd_dct.cpp:1662-1676: the rest stays the caller's)
inline void copy_pieces(uint8_t *dst, size_t dpitch, const uint8_t *src, size_t spitch, size_t len, size_t count, size_t keep, size_t period)
{
  for (size_t p = 0; p < count; p++)
  {
    uint8_t *d = dst + p * dpitch;
    const uint8_t *s = src + p * spitch;
    if (period == 0)
      memcpy(d, s, len);
    else
      for (size_t o = 0; o < len; o += period)
        memcpy(d + o, s + o, len - o < keep ? len - o : keep);
  }
}

// One direction's view of the planes: `count` pieces per block row range, piece p of block rows [r0, r1) at base + p * stride + r0 * row,
// (r1 - r0) * row bytes -- host plane and device mirror alike.  Strip layouts (Q32, BLOCK, BLOCK_SSE): one piece, row = 8 * sizeX.  The
// stereo layout reads 2 pieces (the stacked images, simd_dct.cpp:1089-1099) and writes 64 (the coefficient planes, :1061).
struct Pieces
{
  size_t count, stride, row;
};

// worker k of n takes this part of a chunk's pieces: whole pieces when there are enough of them, else a byte range of every piece
struct Share
{
  size_t p0, p1, o0, o1;
};
inline Share share_of(size_t count, size_t len, size_t k, size_t n, size_t align)
{
  if (count >= 2 * n)
    return {count * k / n, count * (k + 1) / n, 0, len};
  const size_t a = (len * k / n) / align * align, b = k + 1 == n ? len : (len * (k + 1) / n) / align * align;
  return {0, count, a, b};
}

// The host pipeline's extra hands.  Copying between the caller's pageable memory and the pinned bounce buffers must keep up with
// a link that moves ~48 GB/s each way at once (tools/pcie_bench, profiles/r05_pcie_bench.log) while one core sustains ~27-30 GB/s of
// memcpy: helper threads per calling thread (started on first use, joined when the thread's staging is released) take shares of
// every chunk's input copy (one pool) and the output copies -- wait for the chunk's device->host copy, then pinned -> caller memory,
// in two halves -- (another pool: an output job blocks its thread on an event, an input share must never queue behind it).
// A latch per pipeline slot and direction says when a slot's buffers are free again.
template <class Dev>
struct CopyPool
{
  typedef typename Dev::stream_t stream_t;
  typedef typename Dev::event_t event_t;
  struct Job
  {
    bool has_stream;   // wait for `stream` first
    stream_t stream;
    event_t *event;    // (or) wait for this event first; may be null
    uint8_t *dst;
    const uint8_t *src;
    size_t len;        // 0: nothing to copy (the data was DMA'd straight into pinned caller memory)
    std::atomic<int> *latch;
    // several pieces of `len` bytes, `dst_pitch` / `src_pitch` apart (count <= 1: one piece), and of every `period` bytes only the first
    // `keep` (period 0: all of them) -- copy_pieces below
    size_t count = 1, dst_pitch = 0, src_pitch = 0, keep = 0, period = 0;
  };
  enum { kThreads = 3 };
  std::thread th[kThreads];
  int started = 0, wanted = kThreads;
  std::mutex m;
  std::condition_variable cv_job, cv_done;
  std::deque<Job> q;
  bool stop = false;
  std::atomic<bool> failed{false};
  Dev *dev = nullptr;

  void run()
  {
    dev->bind_thread();
    for (;;)
    {
      Job j;
      {
        std::unique_lock<std::mutex> lk(m);
        cv_job.wait(lk, [&] { return stop || !q.empty(); });
        if (q.empty())
          return; // stop, and every queued job has been served
        j = q.front();
        q.pop_front();
      }
      const bool ok = (!j.has_stream || dev->stream_wait(j.stream)) && (!j.event || dev->event_wait(*j.event));
      if (ok && j.len)
        copy_pieces(j.dst, j.dst_pitch, j.src, j.src_pitch, j.len, j.count < 1 ? 1 : j.count, j.keep, j.period);
      if (!ok)
        failed = true;
      {
        std::lock_guard<std::mutex> lk(m); // the waiter checks the latch under this mutex: no lost wake-up
        j.latch->fetch_sub(1);
      }
      cv_done.notify_all();
    }
  }
  bool start(Dev *d, int threads = kThreads)
  {
    wanted = threads < 1 ? 1 : (threads > kThreads ? (int)kThreads : threads);
    if (started == wanted)
    {
      dev = d;
      return true;
    }
    if (started) // a partial start earlier: do without helpers
      return false;
    dev = d;
    stop = false;
    try
    {
      for (; started < wanted; started++)
        th[started] = std::thread([this] { run(); });
    }
    catch (...)
    {
      shutdown();
      return false;
    }
    return true;
  }
  void push(const Job &j)
  {
    {
      std::lock_guard<std::mutex> lk(m);
      q.push_back(j);
    }
    cv_job.notify_one();
  }
  void wait(std::atomic<int> &latch)
  {
    std::unique_lock<std::mutex> lk(m);
    cv_done.wait(lk, [&] { return latch.load() == 0; });
  }
  // queued jobs are served before the helpers leave (their latches reach 0), then the threads are joined
  void shutdown()
  {
    {
      std::lock_guard<std::mutex> lk(m);
      stop = true;
    }
    cv_job.notify_all();
    for (int i = 0; i < started; i++)
      if (th[i].joinable())
        th[i].join();
    started = 0;
    q.clear();
  }
  ~CopyPool() { shutdown(); }
};

enum { PIPELINE_OK = 0, PIPELINE_FAILED = -1 }; // a positive return is the status of a failed launch
enum { kPipeSlots = 4 };                        // chunks in flight per direction

// What the strip pipeline works with: the caller's host planes, the device mirrors of both planes, kPipeSlots pinned bounce
// buffers per direction, ONE STREAM PER STAGE (host->device copies, kernels, device->host copies) with an event per slot and stage
// between them, one latch per slot and direction.
//
// Round 5 (profiles/r05_pcie_bench.log, 32 MiB each way on this box): the link moves 47.9 GB/s each way when both directions run
// at once (0.70 ms), but only 27-32 GB/s in the pattern rounds 2-4 used -- two streams, each carrying its chunk's copy in, kernel
// and copy out one after the other -- and two slots per direction meant two chunks per (copy in + DMA in + kernel + DMA out + copy
// out) latency, ~26 GB/s, whatever the link could do.  Now every stage has its own stream, so each DMA engine sees its copies back
// to back, and four slots keep all five stages busy.
template <class Dev>
struct StripPipeline
{
  typedef typename Dev::stream_t stream_t;
  typedef typename Dev::event_t event_t;
  typedef typename CopyPool<Dev>::Job Job;
  Dev *dev;
  CopyPool<Dev> *pool_in, *pool_out;
  const uint8_t *from; // caller's input plane (host)
  uint8_t *to;         // caller's output plane (host)
  uint8_t *d_in, *d_out;
  uint8_t *const *pin_in, *const *pin_out; // [kPipeSlots] each
  stream_t s_in, s_k, s_out;
  event_t *e_in, *e_k, *e_out;             // [kPipeSlots] each
  std::atomic<int> *in_latch, *out_latch;  // [kPipeSlots] each
  Pieces in, out;        // where the block rows of either plane lie
  size_t rows_per_chunk;
  bool pinned_in, pinned_out; // caller memory is DMA-able in place: no bounce buffer, no memcpy
  bool use_helpers;
  // of every out_period bytes of the output strips only the first out_keep are the call's, the rest stays the caller's (0: all); the caller
  // passes pinned_out = false with it.  out_tail: bytes right behind the last processed row that the call writes too (the SSE encq tier's
  // surviving spill, simd_dct.cpp:1676), handed over after the last chunk.  Single-piece layouts only.
  size_t out_keep = 0, out_period = 0, out_tail = 0;

  // Block rows [b0, b1) in chunks of rows_per_chunk rows; chunk c uses slot c % kPipeSlots of either direction, its pieces packed back
  // to back in the slot's bounce buffer:
  //   caller -> pin_in[slot]         the calling thread and two helpers, a third each (after the slot's previous DMA in has left it)
  //   pin_in[slot] -> device         s_in (one 2-D copy when there are several pieces);  e_in[slot] behind it
  //   kernel                         s_k, after e_in[slot];  e_k[slot] behind it
  //   device -> pin_out[slot]        s_out, after e_k[slot] and after the helpers have emptied the slot;  e_out[slot] behind it
  //   pin_out[slot] -> caller        two helpers, half each, after e_out[slot]
  // On every exit path -- success, a failed copy, a failed launch -- every job handed to the helpers has finished and all three
  // streams are idle: nothing still targets the caller's memory or the bounce buffers when this returns.
  int run(size_t b0, size_t b1)
  {
    const size_t nchunks = ceil_div(b1 - b0, rows_per_chunk);
    const bool helpers = use_helpers && nchunks > 1 && !(pinned_in && pinned_out) && pool_in->start(dev, 2) && pool_out->start(dev, 3);
    if (helpers)
    {
      pool_in->failed = false;
      pool_out->failed = false;
    }
    auto abandon = [&]() { // every job handed to the helpers finishes before the buffers are reused or freed
      if (helpers)
        for (int sl = 0; sl < kPipeSlots; sl++)
        {
          pool_in->wait(in_latch[sl]);
          pool_out->wait(out_latch[sl]);
        }
      (void)dev->stream_wait(s_in);
      (void)dev->stream_wait(s_k);
      (void)dev->stream_wait(s_out);
      return (int)PIPELINE_FAILED;
    };
    // one worker's share of a chunk between the caller's plane (pieces `pc.stride` apart) and a bounce buffer (pieces packed, `len` apart)
    auto job = [](const Share &sh, uint8_t *dst, size_t dpitch, const uint8_t *src, size_t spitch, event_t *ev, std::atomic<int> *latch, size_t keep, size_t period) {
      Job j{false, stream_t(), ev, dst + sh.p0 * dpitch + sh.o0, src + sh.p0 * spitch + sh.o0, sh.o1 - sh.o0, latch};
      j.count = sh.p1 - sh.p0;
      j.dst_pitch = dpitch;
      j.src_pitch = spitch;
      j.keep = keep;
      j.period = period;
      return j;
    };
    auto run_job = [](const Job &j) {
      if (j.len)
        copy_pieces(j.dst, j.dst_pitch, j.src, j.src_pitch, j.len, j.count, j.keep, j.period);
    };
    int r = 0;
    for (size_t c = 0; c < nchunks && r == 0; c++)
    {
      const int sl = (int)(c % kPipeSlots);
      const size_t r0 = b0 + c * rows_per_chunk, r1 = r0 + rows_per_chunk < b1 ? r0 + rows_per_chunk : b1;
      const size_t ioff = r0 * in.row, ilen = (r1 - r0) * in.row; // per piece
      const size_t ooff = r0 * out.row, olen = (r1 - r0) * out.row;
      bool in_place = pinned_in; // pinned caller memory is DMA'd in place
      if (!pinned_in)
      {
        if (c >= (size_t)kPipeSlots && !dev->event_wait(e_in[sl])) // chunk c - kPipeSlots has left the bounce buffer
          return abandon();
        const size_t workers = helpers ? 3 : 1;
        if (helpers)
        {
          in_latch[sl] = 2;
          for (size_t k = 1; k < 3; k++)
            pool_in->push(job(share_of(in.count, ilen, k, 3, 64), pin_in[sl], ilen, from + ioff, in.stride, nullptr, &in_latch[sl], 0, 0));
        }
        run_job(job(share_of(in.count, ilen, 0, workers, 64), pin_in[sl], ilen, from + ioff, in.stride, nullptr, nullptr, 0, 0));
        if (helpers)
          pool_in->wait(in_latch[sl]);
      }
      const bool copied_in = in.count == 1 ? dev->h2d_async(d_in + ioff, in_place ? from + ioff : pin_in[sl], ilen, s_in)
                                           : dev->h2d_2d_async(d_in + ioff, in.stride, in_place ? from + ioff : pin_in[sl], in_place ? in.stride : ilen, ilen, in.count, s_in);
      if (!copied_in || !dev->event_record(e_in[sl], s_in) || !dev->stream_wait_event(s_k, e_in[sl]))
        return abandon();
      r = dev->launch(r0, r1, s_k);
      if (r != 0)
        break;
      if (!dev->event_record(e_k[sl], s_k))
        return abandon();
      if (helpers && !pinned_out && c >= (size_t)kPipeSlots)
      { // the output bounce buffer of this slot: chunk c - kPipeSlots is with the caller
        pool_out->wait(out_latch[sl]);
        if (pool_out->failed.load())
          return abandon();
      }
      if (!dev->stream_wait_event(s_out, e_k[sl]))
        return abandon();
      const bool copied_out = out.count == 1 ? dev->d2h_async(pinned_out ? to + ooff : pin_out[sl], d_out + ooff, olen, s_out)
                                             : dev->d2h_2d_async(pinned_out ? to + ooff : pin_out[sl], pinned_out ? out.stride : olen, d_out + ooff, out.stride, olen, out.count, s_out);
      if (!copied_out)
        return abandon();
      if (!pinned_out)
      {
        if (!dev->event_record(e_out[sl], s_out))
          return abandon();
        const size_t align = out_period ? out_period : 64;
        if (helpers)
        {
          out_latch[sl] = 2;
          for (size_t k = 0; k < 2; k++)
            pool_out->push(job(share_of(out.count, olen, k, 2, align), to + ooff, out.stride, pin_out[sl], olen, &e_out[sl], &out_latch[sl], out_keep, out_period));
        }
        else
        { // without helpers: one chunk at a time on the way out
          if (!dev->event_wait(e_out[sl]))
            return abandon();
          run_job(job(share_of(out.count, olen, 0, 1, align), to + ooff, out.stride, pin_out[sl], olen, nullptr, nullptr, out_keep, out_period));
        }
      }
    }
    if (r != 0)
    {
      (void)abandon();
      return r;
    }
    if (helpers)
    {
      for (int sl = 0; sl < kPipeSlots; sl++)
        pool_out->wait(out_latch[sl]);
      if (pool_out->failed.load() || pool_in->failed.load())
        return abandon();
    }
    if (out_tail && nchunks > 0)
    { // behind the last kernel on s_out's own order (every slot is free again: through slot 0's bounce buffer)
      const size_t toff = b1 * out.row;
      if (!dev->stream_wait(s_k) || !dev->d2h_async(pinned_out ? to + toff : pin_out[0], d_out + toff, out_tail, s_out) || !dev->stream_wait(s_out))
        return abandon();
      if (!pinned_out)
        memcpy(to + toff, pin_out[0], out_tail);
    }
    // s_out's last copy is behind every kernel, which is behind every copy in: when s_out is idle, so is the pipeline
    if (!dev->stream_wait(s_out) || !dev->stream_wait(s_k) || !dev->stream_wait(s_in))
      return abandon();
    return PIPELINE_OK;
  }
};

} // namespace mdct_host
#endif
