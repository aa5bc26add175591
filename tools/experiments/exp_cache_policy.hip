// exp_cache_policy.hip -- the streaming copy (8 x 16 B per lane, 256-thread workgroups: k_stream_copy's shape) with every combination of the
// gfx950 cache-policy bits (sc0, sc1, nt) on its loads and on its stores, against the compiler's __builtin_nontemporal_load/store (nt only).
// Is there a policy under which a read-N/write-N stream exceeds the ~6.3 TB/s every HBM-bound kernel of the engine is measured against?
//   hipcc --offload-arch=gfx950 -O3 tools/experiments/exp_cache_policy.hip -o tools/experiments/exp_cache_policy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

#define KERNEL(name, LMOD, SMOD)                                                                                   \
  __global__ __launch_bounds__(256) void name(const u4 *__restrict__ a, u4 *__restrict__ b)                        \
  {                                                                                                                \
    const size_t base = (size_t)blockIdx.x * 256 * 8 + threadIdx.x;                                                \
    u4 v[8];                                                                                                       \
    _Pragma("unroll") for (int u = 0; u < 8; u++)                                                                  \
    {                                                                                                              \
      const u4 *p = a + base + (size_t)u * 256;                                                                    \
      asm volatile("global_load_dwordx4 %0, %1, off" LMOD : "=&v"(v[u]) : "v"(p) : "memory");                      \
    }                                                                                                              \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                               \
    _Pragma("unroll") for (int u = 0; u < 8; u++)                                                                  \
    {                                                                                                              \
      u4 *q = b + base + (size_t)u * 256;                                                                          \
      asm volatile("global_store_dwordx4 %0, %1, off" SMOD ::"v"(q), "v"(v[u]) : "memory");                        \
    }                                                                                                              \
  }

#define ROW(L, LMOD)                                     \
  KERNEL(k_##L##_p, LMOD, "")                            \
  KERNEL(k_##L##_nt, LMOD, " nt")                        \
  KERNEL(k_##L##_sc0, LMOD, " sc0")                      \
  KERNEL(k_##L##_sc1, LMOD, " sc1")                      \
  KERNEL(k_##L##_sc0sc1, LMOD, " sc0 sc1")               \
  KERNEL(k_##L##_sc0nt, LMOD, " sc0 nt")                 \
  KERNEL(k_##L##_sc1nt, LMOD, " sc1 nt")                 \
  KERNEL(k_##L##_sc0sc1nt, LMOD, " sc0 sc1 nt")
ROW(p, "")
ROW(nt, " nt")
ROW(sc0, " sc0")
ROW(sc1, " sc1")
ROW(sc0sc1, " sc0 sc1")
ROW(sc0nt, " sc0 nt")
ROW(sc1nt, " sc1 nt")
ROW(sc0sc1nt, " sc0 sc1 nt")

__global__ __launch_bounds__(256) void k_builtin_nt(const u4 *__restrict__ a, u4 *__restrict__ b)
{
  const size_t base = (size_t)blockIdx.x * 256 * 8 + threadIdx.x;
  u4 v[8];
#pragma unroll
  for (int u = 0; u < 8; u++)
    v[u] = __builtin_nontemporal_load(a + base + (size_t)u * 256);
#pragma unroll
  for (int u = 0; u < 8; u++)
    __builtin_nontemporal_store(v[u], b + base + (size_t)u * 256);
}

// loads by the compiler (its own incremental s_waitcnt), only the stores spelled out: comparable with k_builtin_nt
#define KSTORE(name, SMOD)                                                                          \
  __global__ __launch_bounds__(256) void name(const u4 *__restrict__ a, u4 *__restrict__ b)         \
  {                                                                                                 \
    const size_t base = (size_t)blockIdx.x * 256 * 8 + threadIdx.x;                                 \
    u4 v[8];                                                                                        \
    _Pragma("unroll") for (int u = 0; u < 8; u++) v[u] = __builtin_nontemporal_load(a + base + (size_t)u * 256); \
    _Pragma("unroll") for (int u = 0; u < 8; u++)                                                   \
    {                                                                                               \
      u4 *q = b + base + (size_t)u * 256;                                                           \
      asm volatile("global_store_dwordx4 %0, %1, off" SMOD ::"v"(q), "v"(v[u]) : "memory");         \
    }                                                                                               \
  }
KSTORE(ks_nt, " nt")
KSTORE(ks_sc1, " sc1")
KSTORE(ks_sc1nt, " sc1 nt")
KSTORE(ks_sc0sc1, " sc0 sc1")
KSTORE(ks_sc0sc1nt, " sc0 sc1 nt")
KSTORE(ks_plain, "")

typedef void (*kern_t)(const u4 *, u4 *);
struct Case
{
  const char *load, *store;
  kern_t k;
};
#define CROW(L, LN)                                                                                                                                    \
  {LN, "plain", k_##L##_p}, {LN, "nt", k_##L##_nt}, {LN, "sc0", k_##L##_sc0}, {LN, "sc1", k_##L##_sc1}, {LN, "sc0 sc1", k_##L##_sc0sc1},              \
      {LN, "sc0 nt", k_##L##_sc0nt}, {LN, "sc1 nt", k_##L##_sc1nt}, {LN, "sc0 sc1 nt", k_##L##_sc0sc1nt},

int main()
{
  const size_t bytes = (size_t)8192 * 8192 * 2; // the bench's plane: 134 MB in, 134 MB out per launch
  const int NS = 4;                             // rotating sets: 1 GB, past the 256 MB Infinity Cache
  std::vector<u4 *> a(NS), b(NS);
  for (int i = 0; i < NS; i++)
  {
    CK(hipMalloc(&a[i], bytes));
    CK(hipMalloc(&b[i], bytes));
    CK(hipMemset(a[i], i + 1, bytes));
    CK(hipMemset(b[i], 0, bytes));
  }
  const unsigned grid = (unsigned)(bytes / 16 / (256 * 8));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  std::vector<Case> cases = {{"builtin nt", "builtin nt", k_builtin_nt}, {"builtin nt", "asm nt", ks_nt}, {"builtin nt", "asm sc1", ks_sc1}, {"builtin nt", "asm sc1 nt", ks_sc1nt},
                             {"builtin nt", "asm sc0 sc1", ks_sc0sc1}, {"builtin nt", "asm sc0sc1nt", ks_sc0sc1nt}, {"builtin nt", "asm plain", ks_plain},
                             {"builtin nt", "builtin nt", k_builtin_nt}, {"builtin nt", "asm sc1 nt", ks_sc1nt}, {"builtin nt", "asm sc0sc1nt", ks_sc0sc1nt}, {"builtin nt", "asm nt", ks_nt},
                             CROW(p, "plain") CROW(nt, "nt") CROW(sc0, "sc0") CROW(sc1, "sc1") CROW(sc0sc1, "sc0 sc1") CROW(sc0nt, "sc0 nt") CROW(sc1nt, "sc1 nt") CROW(sc0sc1nt, "sc0 sc1 nt")};
  printf("%-12s %-12s %9s %9s\n", "loads", "stores", "us", "TB/s");
  for (const Case &c : cases)
  {
    for (int i = 0; i < 40; i++)
      hipLaunchKernelGGL(c.k, dim3(grid), dim3(256), 0, 0, a[i % NS], b[i % NS]);
    std::vector<float> t;
    for (int rep = 0; rep < 7; rep++)
    {
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < 40; i++)
        hipLaunchKernelGGL(c.k, dim3(grid), dim3(256), 0, 0, a[i % NS], b[i % NS]);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      t.push_back(ms / 40);
    }
    std::sort(t.begin(), t.end());
    printf("%-12s %-12s %9.2f %9.3f\n", c.load, c.store, t[3] * 1e3, 2.0 * bytes / (t[3] * 1e-3) / 1e12);
    fflush(stdout);
  }
  // the data really arrived (last case's policy included)
  std::vector<unsigned char> h(64);
  CK(hipMemcpy(h.data(), b[1], 64, hipMemcpyDeviceToHost));
  printf("check: b[1][0] = %u (want 2)\n", h[0]);
  return 0;
}
