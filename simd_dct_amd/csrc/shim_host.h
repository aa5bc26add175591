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
        memcpy(j.dst, j.src, j.len);
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
  Dev *dev;
  CopyPool<Dev> *pool_in, *pool_out;
  const uint8_t *from; // caller's input plane (host)
  uint8_t *to;         // caller's output plane (host)
  uint8_t *d_in, *d_out;
  uint8_t *const *pin_in, *const *pin_out; // [kPipeSlots] each
  stream_t s_in, s_k, s_out;
  event_t *e_in, *e_k, *e_out;             // [kPipeSlots] each
  std::atomic<int> *in_latch, *out_latch;  // [kPipeSlots] each
  size_t strip;          // bytes per block row, input and output alike
  size_t rows_per_chunk;
  bool pinned_in, pinned_out; // caller memory is DMA-able in place: no bounce buffer, no memcpy
  bool use_helpers;

  // Block rows [b0, b1) in chunks of rows_per_chunk rows; chunk c uses slot c % kPipeSlots of either direction:
  //   caller -> pin_in[slot]         the calling thread and two helpers, a third each (after the slot's previous DMA in has left it)
  //   pin_in[slot] -> device         s_in;  e_in[slot] behind it
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
    int r = 0;
    for (size_t c = 0; c < nchunks && r == 0; c++)
    {
      const int sl = (int)(c % kPipeSlots);
      const size_t r0 = b0 + c * rows_per_chunk, r1 = r0 + rows_per_chunk < b1 ? r0 + rows_per_chunk : b1;
      const size_t off = r0 * strip, len = (r1 - r0) * strip;
      const uint8_t *h_in = from + off; // pinned caller memory is DMA'd in place
      if (!pinned_in)
      {
        if (c >= (size_t)kPipeSlots && !dev->event_wait(e_in[sl])) // chunk c - kPipeSlots has left the bounce buffer
          return abandon();
        const size_t third = helpers ? (len / 3) & ~(size_t)63 : 0, mine = len - 2 * third;
        if (helpers)
        {
          in_latch[sl] = 2;
          pool_in->push({false, stream_t(), nullptr, pin_in[sl] + mine, from + off + mine, third, &in_latch[sl]});
          pool_in->push({false, stream_t(), nullptr, pin_in[sl] + mine + third, from + off + mine + third, third, &in_latch[sl]});
        }
        memcpy(pin_in[sl], from + off, mine);
        if (helpers)
          pool_in->wait(in_latch[sl]);
        h_in = pin_in[sl];
      }
      if (!dev->h2d_async(d_in + off, h_in, len, s_in) || !dev->event_record(e_in[sl], s_in) || !dev->stream_wait_event(s_k, e_in[sl]))
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
      if (!dev->stream_wait_event(s_out, e_k[sl]) || !dev->d2h_async(pinned_out ? to + off : pin_out[sl], d_out + off, len, s_out))
        return abandon();
      if (!pinned_out)
      {
        if (!dev->event_record(e_out[sl], s_out))
          return abandon();
        if (helpers)
        {
          const size_t half = (len / 2) & ~(size_t)63;
          out_latch[sl] = 2;
          pool_out->push({false, stream_t(), &e_out[sl], to + off, pin_out[sl], half, &out_latch[sl]});
          pool_out->push({false, stream_t(), &e_out[sl], to + off + half, pin_out[sl] + half, len - half, &out_latch[sl]});
        }
        else
        { // without helpers: one chunk at a time on the way out
          if (!dev->event_wait(e_out[sl]))
            return abandon();
          memcpy(to + off, pin_out[sl], len);
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
    // s_out's last copy is behind every kernel, which is behind every copy in: when s_out is idle, so is the pipeline
    if (!dev->stream_wait(s_out) || !dev->stream_wait(s_k) || !dev->stream_wait(s_in))
      return abandon();
    return PIPELINE_OK;
  }
};

} // namespace mdct_host
#endif
