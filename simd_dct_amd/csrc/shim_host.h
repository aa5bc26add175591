// shim_host.h -- the HOST-ONLY logic of the drop-in shim (csrc/shim.hip), free of any HIP type so that it also builds
// with plain g++: the reference's row-range arithmetic, the tier choice from the reference's CPU-flag globals, the
// helper-thread copy pool and the chunked two-slot strip pipeline that serves host-pointer calls.  shim.hip
// instantiates the templates with the HIP runtime as back end; tests/shim_host_driver.cpp instantiates them with
// worker-thread "streams" over plain memory and runs them under -fsanitize=thread and -fsanitize=address,undefined
// (the GPU pool cannot run sanitizers).
//
// Back end contract (`Dev`):
//   typedef ... stream_t;                                        copyable handle, value-initialisable to "none"
//   void bind_thread();                                          called once by every helper thread before it works
//   bool stream_wait(stream_t);                                  block until everything queued on the stream is done
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

// The host pipeline's extra hands.  Copying between the caller's pageable memory and the pinned bounce buffers is what
// bounds a host-pointer call: one core sustains ~15 GB/s of memcpy, PCIe moves ~26 GB/s each way at once.  Three helper
// threads per calling thread (started on first use, joined when the thread's staging is released) take (a) half of every
// chunk's input copy and (b) the output copies -- wait for the chunk's stream, then pinned -> caller memory, in two halves
// -- while the calling thread copies the next chunk's input.  A latch per pipeline slot and direction says when a
// slot's buffers are free again.
template <class Dev>
struct CopyPool
{
  typedef typename Dev::stream_t stream_t;
  struct Job
  {
    bool has_stream;   // wait for `stream` first
    stream_t stream;
    uint8_t *dst;
    const uint8_t *src;
    size_t len;        // 0: nothing to copy (the data was DMA'd straight into pinned caller memory)
    std::atomic<int> *latch;
  };
  enum { kThreads = 3 };
  std::thread th[kThreads];
  int started = 0;
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
      const bool ok = !j.has_stream || dev->stream_wait(j.stream);
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
  bool start(Dev *d)
  {
    if (started == kThreads)
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
      for (; started < kThreads; started++)
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

// What a two-slot strip pipeline works with: the caller's host planes, the device mirrors of both planes, two pinned
// bounce buffers per direction, two streams, one latch per slot and direction.
template <class Dev>
struct StripPipeline
{
  typedef typename Dev::stream_t stream_t;
  Dev *dev;
  CopyPool<Dev> *pool;
  const uint8_t *from; // caller's input plane (host)
  uint8_t *to;         // caller's output plane (host)
  uint8_t *d_in, *d_out;
  uint8_t *pin_in[2], *pin_out[2];
  stream_t stream[2];
  std::atomic<int> *in_latch, *out_latch; // [2] each
  size_t strip;          // bytes per block row, input and output alike
  size_t rows_per_chunk;
  bool pinned_in, pinned_out; // caller memory is DMA-able in place: no bounce buffer, no memcpy
  bool use_helpers;

  // Block rows [b0, b1): strips of rows_per_chunk rows ping-pong over the two slots, so strip k's kernel and
  // device->host copy overlap strip k+1's host->device copy; the caller's pageable memory is touched only by plain
  // memcpy to/from the bounce buffers.  On every exit path -- success, a failed copy, a failed launch -- every job
  // handed to the helpers has finished and (failure paths) both streams are idle: nothing still targets the
  // caller's memory or the bounce buffers when this returns.
  int run(size_t b0, size_t b1)
  {
    const size_t nchunks = ceil_div(b1 - b0, rows_per_chunk);
    const bool helpers = use_helpers && nchunks > 1 && !(pinned_in && pinned_out) && pool->start(dev);
    if (helpers)
      pool->failed = false;
    auto chunk_rows = [&](size_t c, size_t *r0, size_t *r1) {
      *r0 = b0 + c * rows_per_chunk;
      *r1 = *r0 + rows_per_chunk < b1 ? *r0 + rows_per_chunk : b1;
    };
    auto drain = [&](size_t c) { // chunk c has left both bounce buffers of its slot; its output is with the caller
      const int sl = (int)(c & 1);
      if (helpers)
      {
        pool->wait(out_latch[sl]);
        return !pool->failed.load();
      }
      size_t r0, r1;
      chunk_rows(c, &r0, &r1);
      if (!dev->stream_wait(stream[sl]))
        return false;
      if (!pinned_out)
        memcpy(to + r0 * strip, pin_out[sl], (r1 - r0) * strip);
      return true;
    };
    auto abandon = [&]() { // every job handed to the helpers finishes before the buffers are reused or freed
      if (helpers)
        for (int sl = 0; sl < 2; sl++)
        {
          pool->wait(in_latch[sl]);
          pool->wait(out_latch[sl]);
        }
      for (int sl = 0; sl < 2; sl++)
        (void)dev->stream_wait(stream[sl]);
      return (int)PIPELINE_FAILED;
    };
    int r = 0;
    for (size_t c = 0; c < nchunks && r == 0; c++)
    {
      const int sl = (int)(c & 1);
      if (c >= 2 && !drain(c - 2))
        return abandon();
      size_t r0, r1;
      chunk_rows(c, &r0, &r1);
      const size_t off = r0 * strip, len = (r1 - r0) * strip;
      const uint8_t *h_in = from + off; // pinned caller memory is DMA'd in place
      if (!pinned_in)
      {
        const size_t mine = helpers ? (len / 2) & ~(size_t)63 : len; // a helper copies the rest meanwhile
        if (helpers)
        {
          in_latch[sl] = 1;
          pool->push({false, stream_t(), pin_in[sl] + mine, from + off + mine, len - mine, &in_latch[sl]});
        }
        memcpy(pin_in[sl], from + off, mine);
        if (helpers)
          pool->wait(in_latch[sl]);
        h_in = pin_in[sl];
      }
      if (!dev->h2d_async(d_in + off, h_in, len, stream[sl]))
        return abandon();
      r = dev->launch(r0, r1, stream[sl]);
      if (r == 0 && !dev->d2h_async(pinned_out ? to + off : pin_out[sl], d_out + off, len, stream[sl]))
        return abandon();
      if (helpers)
      { // queued even when the launch failed: the slot's earlier copies still have to be waited for
        const size_t out_len = pinned_out || r != 0 ? 0 : len, half = (out_len / 2) & ~(size_t)63;
        out_latch[sl] = 2;
        pool->push({true, stream[sl], to + off, pin_out[sl], half, &out_latch[sl]});
        pool->push({true, stream[sl], to + off + half, pin_out[sl] + half, out_len - half, &out_latch[sl]});
      }
    }
    if (r != 0)
    {
      (void)abandon();
      return r;
    }
    for (size_t c = nchunks >= 2 ? nchunks - 2 : 0; c < nchunks; c++)
      if (!drain(c))
        return abandon();
    return PIPELINE_OK;
  }
};

} // namespace mdct_host
#endif
