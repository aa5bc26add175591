// membench.hip -- what "HBM roofline" means on this box: read-N/write-N copies in several
// shapes (the engine's own access shapes included), timed with HIP events.
// Build: hipcc --offload-arch=gfx950 -O3 tools/membench.hip -o tools/membench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void copy_gridstride(const uint4 *__restrict__ a, uint4 *__restrict__ b, size_t n)
{
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
    b[i] = a[i];
}
template <int U>
__global__ __launch_bounds__(256) void copy_oneshot(const uint4 *__restrict__ a, uint4 *__restrict__ b, size_t n)
{
  const size_t base = (size_t)blockIdx.x * 256 * U + threadIdx.x;
  uint4 v[U];
#pragma unroll
  for (int u = 0; u < U; u++)
    v[u] = a[base + (size_t)u * 256];
#pragma unroll
  for (int u = 0; u < U; u++)
    b[base + (size_t)u * 256] = v[u];
}
template <int U>
__global__ __launch_bounds__(256) void copy_oneshot_nt(const uint4 *__restrict__ a, uint4 *__restrict__ b, size_t n)
{
  const size_t base = (size_t)blockIdx.x * 256 * U + threadIdx.x;
  typedef unsigned int u4 __attribute__((ext_vector_type(4)));
  const u4 *a4 = reinterpret_cast<const u4 *>(a);
  u4 *b4 = reinterpret_cast<u4 *>(b);
  u4 v[U];
#pragma unroll
  for (int u = 0; u < U; u++)
    v[u] = __builtin_nontemporal_load(a4 + base + (size_t)u * 256);
#pragma unroll
  for (int u = 0; u < U; u++)
    __builtin_nontemporal_store(v[u], b4 + base + (size_t)u * 256);
}
// the int16 kernels' shape: a lane owns 8 rows x 16 B of an 8192-wide int16 plane
__global__ __launch_bounds__(256) void copy_block_rows(const uint4 *__restrict__ a, uint4 *__restrict__ b, size_t pitch16, unsigned bpr)
{
  const unsigned t = blockIdx.x * 256 + threadIdx.x;
  const unsigned row = t / bpr, bx = t - row * bpr;
  const size_t base = (size_t)row * 8 * pitch16 + bx;
  uint4 v[8];
#pragma unroll
  for (int r = 0; r < 8; r++)
    v[r] = a[base + r * pitch16];
#pragma unroll
  for (int r = 0; r < 8; r++)
    b[base + r * pitch16] = v[r];
}
__global__ __launch_bounds__(256) void read_only(const uint4 *__restrict__ a, uint4 *__restrict__ b, size_t n)
{
  const size_t base = (size_t)blockIdx.x * 256 * 8 + threadIdx.x;
  uint4 acc = make_uint4(0, 0, 0, 0);
#pragma unroll
  for (int u = 0; u < 8; u++)
  {
    const uint4 v = a[base + (size_t)u * 256];
    acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u)
    b[base] = acc;
}
__global__ __launch_bounds__(256) void write_only(const uint4 *__restrict__ a, uint4 *__restrict__ b, size_t n)
{
  const size_t base = (size_t)blockIdx.x * 256 * 8 + threadIdx.x;
#pragma unroll
  for (int u = 0; u < 8; u++)
    b[base + (size_t)u * 256] = make_uint4(u, blockIdx.x, threadIdx.x, 7);
}

int main()
{
  const size_t bytes = 128ull << 20; // one 8192x8192 int16 plane
  const size_t n = bytes / 16;
  const int NS = 4;
  std::vector<uint4 *> A(NS), B(NS);
  for (int i = 0; i < NS; i++)
  {
    CK(hipMalloc(&A[i], bytes));
    CK(hipMalloc(&B[i], bytes));
    CK(hipMemset(A[i], i + 1, bytes));
    CK(hipMemset(B[i], 0, bytes));
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto time = [&](const char *name, auto launch, double bytes_moved) {
    for (int i = 0; i < 8; i++) launch(i % NS);
    hipDeviceSynchronize();
    float best = 1e9f, sum = 0;
    for (int rep = 0; rep < 5; rep++)
    {
      hipEventRecord(e0, 0);
      for (int i = 0; i < 40; i++) launch(i % NS);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      ms /= 40;
      best = ms < best ? ms : best;
      sum += ms;
    }
    printf("%-34s best %7.2f us  %7.1f GB/s   mean %7.2f us\n", name, best * 1e3, bytes_moved / (best * 1e-3) / 1e9, sum / 5 * 1e3);
  };
  for (int g : {1024, 2048, 4096, 8192, 16384})
  {
    char nm[64];
    snprintf(nm, sizeof nm, "gridstride grid=%d", g);
    time(nm, [&](int s) { hipLaunchKernelGGL(copy_gridstride, dim3(g), dim3(256), 0, 0, A[s], B[s], n); }, 2.0 * bytes);
  }
  time("oneshot U=1", [&](int s) { hipLaunchKernelGGL(copy_oneshot<1>, dim3(n / 256), dim3(256), 0, 0, A[s], B[s], n); }, 2.0 * bytes);
  time("oneshot U=2", [&](int s) { hipLaunchKernelGGL(copy_oneshot<2>, dim3(n / 512), dim3(256), 0, 0, A[s], B[s], n); }, 2.0 * bytes);
  time("oneshot U=4", [&](int s) { hipLaunchKernelGGL(copy_oneshot<4>, dim3(n / 1024), dim3(256), 0, 0, A[s], B[s], n); }, 2.0 * bytes);
  time("oneshot U=8", [&](int s) { hipLaunchKernelGGL(copy_oneshot<8>, dim3(n / 2048), dim3(256), 0, 0, A[s], B[s], n); }, 2.0 * bytes);
  time("oneshot nontemporal U=4", [&](int s) { hipLaunchKernelGGL(copy_oneshot_nt<4>, dim3(n / 1024), dim3(256), 0, 0, A[s], B[s], n); }, 2.0 * bytes);
  time("oneshot nontemporal U=8", [&](int s) { hipLaunchKernelGGL(copy_oneshot_nt<8>, dim3(n / 2048), dim3(256), 0, 0, A[s], B[s], n); }, 2.0 * bytes);
  time("block rows 8x16B (i16 shape)", [&](int s) { hipLaunchKernelGGL(copy_block_rows, dim3(8192 / 8 * 1024 / 256), dim3(256), 0, 0, A[s], B[s], (size_t)1024, 1024u); }, 2.0 * bytes);
  time("read only U=8", [&](int s) { hipLaunchKernelGGL(read_only, dim3(n / 2048), dim3(256), 0, 0, A[s], B[s], n); }, 1.0 * bytes);
  time("write only U=8", [&](int s) { hipLaunchKernelGGL(write_only, dim3(n / 2048), dim3(256), 0, 0, A[s], B[s], n); }, 1.0 * bytes);
  time("hipMemcpyDtoD", [&](int s) { hipMemcpyAsync(B[s], A[s], bytes, hipMemcpyDeviceToDevice, 0); }, 2.0 * bytes);
  return 0;
}
