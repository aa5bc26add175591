// pcie_bench.cpp -- what the host <-> HBM link of this box gives a caller of the drop-in API with HOST pointers:
// pinned hipMemcpyAsync host->device alone, device->host alone, both directions at once on two streams; the same from pageable
// memory (what the reference's main.cpp passes: malloc'ed planes, main.cpp:475-493); and the host's own memcpy rate with 1, 2 and 4
// threads (the bounce-buffer copies of the shim's strip pipeline, csrc/shim_host.h).  The ceiling the host-pointer pipeline is
// measured against in tools/simd_dct_cli and INTEGRATION.md 1.
//   hipcc -O2 -std=c++17 tools/pcie_bench.cpp -o tools/pcie_bench && tools/pcie_bench [MiB = 64]
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x)                                                                          \
  do                                                                                   \
  {                                                                                    \
    hipError_t e_ = (x);                                                               \
    if (e_ != hipSuccess)                                                              \
    {                                                                                  \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(1);                                                                         \
    }                                                                                  \
  } while (0)

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <class F>
static double best_of(int reps, F f)
{
  double best = 1e30;
  for (int i = 0; i < reps; i++)
  {
    const double t0 = now();
    f();
    const double dt = now() - t0;
    best = dt < best ? dt : best;
  }
  return best;
}

int main(int argc, char **argv)
{
  const size_t mib = argc > 1 ? (size_t)atoi(argv[1]) : 64, n = mib << 20;
  uint8_t *d_a, *d_b, *p_a, *p_b;
  CK(hipMalloc((void **)&d_a, n));
  CK(hipMalloc((void **)&d_b, n));
  CK(hipHostMalloc((void **)&p_a, n, hipHostMallocDefault));
  CK(hipHostMalloc((void **)&p_b, n, hipHostMallocDefault));
  uint8_t *m_a = (uint8_t *)malloc(n), *m_b = (uint8_t *)malloc(n);
  memset(p_a, 1, n);
  memset(p_b, 2, n);
  memset(m_a, 3, n);
  memset(m_b, 4, n);
  hipStream_t s0, s1;
  CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  const double gb = (double)n / 1e9;
  auto h2d = [&] { CK(hipMemcpyAsync(d_a, p_a, n, hipMemcpyHostToDevice, s0)); CK(hipStreamSynchronize(s0)); };
  auto d2h = [&] { CK(hipMemcpyAsync(p_b, d_b, n, hipMemcpyDeviceToHost, s1)); CK(hipStreamSynchronize(s1)); };
  auto both = [&] {
    CK(hipMemcpyAsync(d_a, p_a, n, hipMemcpyHostToDevice, s0));
    CK(hipMemcpyAsync(p_b, d_b, n, hipMemcpyDeviceToHost, s1));
    CK(hipStreamSynchronize(s0));
    CK(hipStreamSynchronize(s1));
  };
  h2d(); d2h(); both();
  const double t_h2d = best_of(10, h2d), t_d2h = best_of(10, d2h), t_both = best_of(10, both);
  printf("pinned, %zu MiB:   host->device %6.1f GB/s   device->host %6.1f GB/s   both at once %6.1f + %6.1f GB/s (%.3f ms for %zu MiB each way)\n", mib, gb / t_h2d, gb / t_d2h,
         gb / t_both, gb / t_both, t_both * 1e3, mib);
  // chunked the way the strip pipeline issues them: 4 MiB pieces alternating over the two streams, in + out per piece on the same stream
  for (size_t piece : {(size_t)1 << 20, (size_t)4 << 20, (size_t)16 << 20})
  {
    auto chunked = [&] {
      for (size_t o = 0, k = 0; o < n; o += piece, k++)
      {
        hipStream_t s = (k & 1) ? s1 : s0;
        CK(hipMemcpyAsync(d_a + o, p_a + o, piece, hipMemcpyHostToDevice, s));
        CK(hipMemcpyAsync(p_b + o, d_b + o, piece, hipMemcpyDeviceToHost, s));
      }
      CK(hipStreamSynchronize(s0));
      CK(hipStreamSynchronize(s1));
    };
    chunked();
    const double t = best_of(10, chunked);
    printf("pinned, in + out per %2zu MiB piece on alternating streams (the strip pipeline's pattern): %6.1f GB/s each way (%.3f ms)\n", piece >> 20, gb / t, t * 1e3);
  }
  // one stream per direction, chunk k's copy out ordered behind chunk k's copy in by an event (what a three-stream pipeline does;
  // a kernel would sit between the two)
  {
    std::vector<hipEvent_t> ev(n >> 20);
    for (auto &e : ev)
      CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (size_t piece : {(size_t)1 << 20, (size_t)2 << 20, (size_t)4 << 20, (size_t)8 << 20})
    {
      auto dep = [&] {
        for (size_t o = 0, k = 0; o < n; o += piece, k++)
        {
          CK(hipMemcpyAsync(d_a + o, p_a + o, piece, hipMemcpyHostToDevice, s0));
          CK(hipEventRecord(ev[k], s0));
          CK(hipStreamWaitEvent(s1, ev[k], 0));
          CK(hipMemcpyAsync(p_b + o, d_a + o, piece, hipMemcpyDeviceToHost, s1));
        }
        CK(hipStreamSynchronize(s0));
        CK(hipStreamSynchronize(s1));
      };
      dep();
      const double t = best_of(10, dep);
      printf("pinned, one stream per direction, %zu MiB pieces, out(k) behind in(k) by an event: %6.1f GB/s each way (%.3f ms)\n", piece >> 20, gb / t, t * 1e3);
    }
  }
  // two streams per direction, pieces alternating between them: does the engine's per-copy set-up overlap the previous copy?
  {
    hipStream_t si[2] = {s0, nullptr}, so[2] = {s1, nullptr};
    CK(hipStreamCreateWithFlags(&si[1], hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&so[1], hipStreamNonBlocking));
    std::vector<hipEvent_t> ev(n >> 20);
    for (auto &e : ev)
      CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (size_t piece : {(size_t)2 << 20, (size_t)4 << 20})
    {
      auto dep = [&] {
        for (size_t o = 0, k = 0; o < n; o += piece, k++)
        {
          CK(hipMemcpyAsync(d_a + o, p_a + o, piece, hipMemcpyHostToDevice, si[k & 1]));
          CK(hipEventRecord(ev[k], si[k & 1]));
          CK(hipStreamWaitEvent(so[k & 1], ev[k], 0));
          CK(hipMemcpyAsync(p_b + o, d_a + o, piece, hipMemcpyDeviceToHost, so[k & 1]));
        }
        for (int i = 0; i < 2; i++)
        {
          CK(hipStreamSynchronize(si[i]));
          CK(hipStreamSynchronize(so[i]));
        }
      };
      dep();
      const double t = best_of(10, dep);
      printf("pinned, TWO streams per direction, %zu MiB pieces alternating, out(k) behind in(k): %6.1f GB/s each way (%.3f ms)\n", piece >> 20, gb / t, t * 1e3);
    }
  }
  // the same from PAGEABLE memory, one host thread per direction (hipMemcpyAsync on pageable memory occupies its calling thread)
  for (size_t piece : {(size_t)1 << 20, (size_t)2 << 20, (size_t)4 << 20, (size_t)8 << 20, n})
  {
    auto two = [&] {
      std::thread tin([&] {
        for (size_t o = 0; o < n; o += piece)
          CK(hipMemcpyAsync(d_a + o, m_a + o, piece, hipMemcpyHostToDevice, s0));
        CK(hipStreamSynchronize(s0));
      });
      std::thread tout([&] {
        for (size_t o = 0; o < n; o += piece)
          CK(hipMemcpyAsync(m_b + o, d_b + o, piece, hipMemcpyDeviceToHost, s1));
        CK(hipStreamSynchronize(s1));
      });
      tin.join();
      tout.join();
    };
    two();
    const double t = best_of(8, two);
    printf("pageable, one host thread per direction, %3zu MiB pieces, both directions at once: %6.1f GB/s each way (%.3f ms)\n", piece >> 20, gb / t, t * 1e3);
  }
  // pageable memory handed straight to hipMemcpyAsync (the runtime pins it on the fly), one host thread per direction, copy out k ordered
  // behind copy in k by an event the first thread records and the second waits for (host handshake, then hipStreamWaitEvent): what a
  // pipeline WITHOUT bounce buffers could do for a caller like main.cpp
  {
    std::vector<hipEvent_t> ev(n >> 20);
    for (auto &e : ev)
      CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (size_t piece : {(size_t)2 << 20, (size_t)4 << 20, (size_t)8 << 20})
    {
      auto dep2 = [&] {
        const size_t cnt = n / piece;
        std::atomic<size_t> recorded{0};
        std::thread tin([&] {
          for (size_t k = 0; k < cnt; k++)
          {
            CK(hipMemcpyAsync(d_a + k * piece, m_a + k * piece, piece, hipMemcpyHostToDevice, s0));
            CK(hipEventRecord(ev[k], s0));
            recorded.store(k + 1, std::memory_order_release);
          }
          CK(hipStreamSynchronize(s0));
        });
        std::thread tout([&] {
          for (size_t k = 0; k < cnt; k++)
          {
            while (recorded.load(std::memory_order_acquire) <= k)
              ;
            CK(hipStreamWaitEvent(s1, ev[k], 0));
            CK(hipMemcpyAsync(m_b + k * piece, d_a + k * piece, piece, hipMemcpyDeviceToHost, s1));
          }
          CK(hipStreamSynchronize(s1));
        });
        tin.join();
        tout.join();
      };
      dep2();
      const double t = best_of(8, dep2);
      printf("pageable DIRECT, one host thread per direction, %zu MiB pieces, out(k) behind in(k): %6.1f GB/s each way (%.3f ms)\n", piece >> 20, gb / t, t * 1e3);
    }
  }
  auto ph2d = [&] { CK(hipMemcpy(d_a, m_a, n, hipMemcpyHostToDevice)); };
  auto pd2h = [&] { CK(hipMemcpy(m_b, d_b, n, hipMemcpyDeviceToHost)); };
  ph2d(); pd2h();
  printf("pageable (malloc), hipMemcpy: host->device %6.1f GB/s   device->host %6.1f GB/s\n", gb / best_of(6, ph2d), gb / best_of(6, pd2h));
  for (int nt : {1, 2, 4, 8})
  {
    auto cp = [&] {
      std::vector<std::thread> th;
      for (int i = 0; i < nt; i++)
        th.emplace_back([&, i] { memcpy(p_a + n / nt * i, m_a + n / nt * i, n / nt); });
      for (auto &t : th)
        t.join();
    };
    cp();
    printf("host memcpy pageable -> pinned, %d thread%s: %6.1f GB/s\n", nt, nt > 1 ? "s" : " ", gb / best_of(8, cp));
  }
  return 0;
}
