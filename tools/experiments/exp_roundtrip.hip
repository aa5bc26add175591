// exp_roundtrip.hip -- A/B harness for variants of the fused int16 round-trip kernel
// (one process, interleaved rounds).  Not part of the product.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -std=c++17 -Iinclude -Isimd_dct_amd/csrc tools/exp_roundtrip.hip -o tools/exp_roundtrip
#include "../simd_dct_amd/csrc/mdct_kernels.hip"
#include <cstdio>
#include <functional>
#include <vector>
#include <algorithm>

using namespace mdct;
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

template <bool NT>
__device__ __forceinline__ uint4 ld16(const void *p)
{
  if constexpr (NT)
  {
    const u4 v = __builtin_nontemporal_load(reinterpret_cast<const u4 *>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
  }
  else
    return *reinterpret_cast<const uint4 *>(p);
}
template <bool NT>
__device__ __forceinline__ void st16(void *p, uint4 v)
{
  if constexpr (NT)
  {
    u4 w = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(w, reinterpret_cast<u4 *>(p));
  }
  else
    *reinterpret_cast<uint4 *>(p) = v;
}

__device__ __forceinline__ void rt_compute(const DctConsts &C, const uint4 (&in)[8], uint4 (&out)[8])
{
  float b[8][8];
#pragma unroll
  for (int r = 0; r < 8; r++)
    unpack_i16x8(in[r], b[r]);
  raw_fwd(C, b);
  raw_inv(C, b);
#pragma unroll
  for (int r = 0; r < 8; r++)
  {
    uint32_t t[8];
#pragma unroll
    for (int c = 0; c < 8; c++)
      t[c] = rne_i16_bits<6>(C, b[r][c]);
    out[r] = make_uint4(pack_lo16(t[0], t[1]), pack_lo16(t[2], t[3]), pack_lo16(t[4], t[5]), pack_lo16(t[6], t[7]));
  }
}

// V1: one block per thread, optional nontemporal, WG size param
template <bool NT, int WG>
__global__ __launch_bounds__(WG) void v_one(I16Args a)
{
  const uint32_t t = blockIdx.x * WG + threadIdx.x;
  if (t >= a.nblocks) return;
  const uint32_t row = t / a.bpr, bx = t - row * a.bpr;
  const int16_t *src = a.from + (size_t)row * 8 * a.pitch_in + (size_t)bx * 8;
  int16_t *dst = a.to + (size_t)row * 8 * a.pitch_out + (size_t)bx * 8;
  uint4 in[8], out[8];
#pragma unroll
  for (int r = 0; r < 8; r++) in[r] = ld16<NT>(src + (size_t)r * a.pitch_in);
  rt_compute(a.consts, in, out);
#pragma unroll
  for (int r = 0; r < 8; r++) st16<NT>(dst + (size_t)r * a.pitch_out, out[r]);
}

// V2: NB blocks per thread (stride = grid*WG blocks), next block's rows prefetched before compute
template <bool NT, int NB>
__global__ __launch_bounds__(256) void v_multi(I16Args a)
{
  const uint32_t stride = gridDim.x * 256;
  uint32_t t = blockIdx.x * 256 + threadIdx.x;
  uint4 in[8], nxt[8], out[8];
  {
    const uint32_t row = t / a.bpr, bx = t - row * a.bpr;
    const int16_t *src = a.from + (size_t)row * 8 * a.pitch_in + (size_t)bx * 8;
#pragma unroll
    for (int r = 0; r < 8; r++) in[r] = ld16<NT>(src + (size_t)r * a.pitch_in);
  }
#pragma unroll 1
  for (int i = 0; i < NB; i++)
  {
    const uint32_t tn = t + stride;
    if (i + 1 < NB)
    {
      const uint32_t row = tn / a.bpr, bx = tn - row * a.bpr;
      const int16_t *src = a.from + (size_t)row * 8 * a.pitch_in + (size_t)bx * 8;
#pragma unroll
      for (int r = 0; r < 8; r++) nxt[r] = ld16<NT>(src + (size_t)r * a.pitch_in);
    }
    rt_compute(a.consts, in, out);
    {
      const uint32_t row = t / a.bpr, bx = t - row * a.bpr;
      int16_t *dst = a.to + (size_t)row * 8 * a.pitch_out + (size_t)bx * 8;
#pragma unroll
      for (int r = 0; r < 8; r++) st16<NT>(dst + (size_t)r * a.pitch_out, out[r]);
    }
#pragma unroll
    for (int r = 0; r < 8; r++) in[r] = nxt[r];
    t = tn;
  }
}

// V3: persistent waves, next tile prefetched by LDS-DMA (global_load_lds_dwordx4, no VGPRs)
// while the current tile is computed.  One wave-private 8 KB LDS tile [8 rows][64 lanes x 16 B].
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

template <int WAVES_PER_WG>
__global__ __launch_bounds__(WAVES_PER_WG * 64) void v_dma(I16Args a, uint32_t ntiles)
{
  __shared__ __attribute__((aligned(16))) uint8_t lds[WAVES_PER_WG][8 * 1024];
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t wave = threadIdx.x >> 6;
  uint8_t *tile = lds[wave];
  const uint32_t total_waves = gridDim.x * WAVES_PER_WG;
  uint32_t t = blockIdx.x * WAVES_PER_WG + wave; // tile index = 64 consecutive blocks
  if (t >= ntiles) return;

  auto issue = [&](uint32_t tt) {
    const uint32_t blk = tt * 64 + lane;
    const uint32_t row = blk / a.bpr, bx = blk - row * a.bpr;
    const int16_t *src = a.from + (size_t)row * 8 * a.pitch_in + (size_t)bx * 8;
#pragma unroll
    for (int r = 0; r < 8; r++)
      __builtin_amdgcn_global_load_lds((gbl_void *)(src + (size_t)r * a.pitch_in), (lds_void *)(tile + r * 1024), 16, 0, 0);
  };
  issue(t);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (;;)
  {
    const uint32_t tn = t + total_waves;
    const bool more = tn < ntiles; // wave-uniform
    uint4 in[8], out[8];
    {
      // LDS reads in asm: hipcc otherwise orders every ds_read behind vmcnt(0) while an LDS-DMA
      // is (or may be) outstanding, which would also wait for the previous tile's stores
      const uint32_t laddr = (uint32_t)(uintptr_t)(lds_void *)(tile) + lane * 16;
      u4 t0, t1, t2, t3, t4, t5, t6, t7;
      asm volatile("ds_read_b128 %0, %8\n ds_read_b128 %1, %8 offset:1024\n ds_read_b128 %2, %8 offset:2048\n ds_read_b128 %3, %8 offset:3072\n"
                   "ds_read_b128 %4, %8 offset:4096\n ds_read_b128 %5, %8 offset:5120\n ds_read_b128 %6, %8 offset:6144\n ds_read_b128 %7, %8 offset:7168\n"
                   "s_waitcnt lgkmcnt(0)"
                   : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7)
                   : "v"(laddr)
                   : "memory");
      __builtin_amdgcn_sched_barrier(0);
      in[0] = make_uint4(t0.x, t0.y, t0.z, t0.w); in[1] = make_uint4(t1.x, t1.y, t1.z, t1.w);
      in[2] = make_uint4(t2.x, t2.y, t2.z, t2.w); in[3] = make_uint4(t3.x, t3.y, t3.z, t3.w);
      in[4] = make_uint4(t4.x, t4.y, t4.z, t4.w); in[5] = make_uint4(t5.x, t5.y, t5.z, t5.w);
      in[6] = make_uint4(t6.x, t6.y, t6.z, t6.w); in[7] = make_uint4(t7.x, t7.y, t7.z, t7.w);
    }
    if (more)
      issue(tn);
    rt_compute(a.consts, in, out);
    {
      const uint32_t blk = t * 64 + lane;
      const uint32_t row = blk / a.bpr, bx = blk - row * a.bpr;
      int16_t *dst = a.to + (size_t)row * 8 * a.pitch_out + (size_t)bx * 8;
#pragma unroll
      for (int r = 0; r < 8; r++) st16<true>(dst + (size_t)r * a.pitch_out, out[r]);
    }
    if (!more) break;
    // the 8 DMA pieces of the next tile are older than this tile's 8 stores: wait for all but
    // the 8 youngest vector-memory ops, i.e. for the DMA only; the stores stay in flight
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    t = tn;
  }
}

// compute only: load once, REPS x (fwd+inv) in registers, store once -> pure issue-rate cost
template <int REPS>
__global__ __launch_bounds__(256) void v_compute(I16Args a)
{
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  if (t >= a.nblocks) return;
  const uint32_t row = t / a.bpr, bx = t - row * a.bpr;
  const int16_t *src = a.from + (size_t)row * 8 * a.pitch_in + (size_t)bx * 8;
  int16_t *dst = a.to + (size_t)row * 8 * a.pitch_out + (size_t)bx * 8;
  uint4 in[8], out[8];
#pragma unroll
  for (int r = 0; r < 8; r++) in[r] = ld16<false>(src + (size_t)r * a.pitch_in);
#pragma unroll 1
  for (int i = 0; i < REPS; i++)
  {
    rt_compute(a.consts, in, out);
#pragma unroll
    for (int r = 0; r < 8; r++) in[r] = out[r];
  }
#pragma unroll
  for (int r = 0; r < 8; r++) st16<false>(dst + (size_t)r * a.pitch_out, out[r]);
}

int main()
{
  const size_t W = 8192, H = 8192, bytes = W * H * 2;
  const int NS = 4;
  std::vector<int16_t *> A(NS), B(NS);
  std::vector<int16_t> host(W * H);
  for (size_t i = 0; i < W * H; i++) host[i] = (int16_t)((i * 2654435761u >> 20) & 0xFF) - 128;
  for (int i = 0; i < NS; i++)
  {
    hipMalloc(&A[i], bytes);
    hipMalloc(&B[i], bytes);
    hipMemcpy(A[i], host.data(), bytes, hipMemcpyHostToDevice);
  }
  I16Args a;
  a.pitch_in = a.pitch_out = W;
  a.bpr = W / 8;
  a.by0 = 0;
  a.nblocks = (uint32_t)(W / 8 * H / 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  struct V { const char *name; std::function<void(int)> f; std::vector<float> t; };
  std::vector<V> vs;
  auto args = [&](int s) { I16Args x = a; x.from = A[s]; x.to = B[s]; return x; };
  vs.push_back({"product k_i16<RT>", [&](int s) { launch_i16(args(s), MODE_ROUNDTRIP, false, 0); }, {}});
  vs.push_back({"one plain WG256", [&](int s) { hipLaunchKernelGGL((v_one<false, 256>), dim3(a.nblocks / 256), dim3(256), 0, 0, args(s)); }, {}});
  vs.push_back({"one NT WG256", [&](int s) { hipLaunchKernelGGL((v_one<true, 256>), dim3(a.nblocks / 256), dim3(256), 0, 0, args(s)); }, {}});
  vs.push_back({"one NT WG64", [&](int s) { hipLaunchKernelGGL((v_one<true, 64>), dim3(a.nblocks / 64), dim3(64), 0, 0, args(s)); }, {}});
  vs.push_back({"one NT WG128", [&](int s) { hipLaunchKernelGGL((v_one<true, 128>), dim3(a.nblocks / 128), dim3(128), 0, 0, args(s)); }, {}});
  vs.push_back({"one NT WG512", [&](int s) { hipLaunchKernelGGL((v_one<true, 512>), dim3(a.nblocks / 512), dim3(512), 0, 0, args(s)); }, {}});
  vs.push_back({"multi2 NT prefetch", [&](int s) { hipLaunchKernelGGL((v_multi<true, 2>), dim3(a.nblocks / 256 / 2), dim3(256), 0, 0, args(s)); }, {}});
  vs.push_back({"multi4 NT prefetch", [&](int s) { hipLaunchKernelGGL((v_multi<true, 4>), dim3(a.nblocks / 256 / 4), dim3(256), 0, 0, args(s)); }, {}});
  vs.push_back({"multi4 plain prefetch", [&](int s) { hipLaunchKernelGGL((v_multi<false, 4>), dim3(a.nblocks / 256 / 4), dim3(256), 0, 0, args(s)); }, {}});
  vs.push_back({"multi8 NT prefetch", [&](int s) { hipLaunchKernelGGL((v_multi<true, 8>), dim3(a.nblocks / 256 / 8), dim3(256), 0, 0, args(s)); }, {}});
  // occupancy limited by reserving (unused) dynamic LDS: 160 KiB per CU / reservation = workgroups per CU
  for (int kb : {0, 32, 40, 53, 80})
  {
    static char names[5][40];
    static int idx = 0;
    snprintf(names[idx], sizeof names[idx], "product RT, %d KiB LDS pad", kb);
    const char *nm = names[idx++];
    vs.push_back({nm, [&, kb](int s) { hipLaunchKernelGGL((k_i16<MODE_ROUNDTRIP, false>), dim3(a.nblocks / 256), dim3(256), kb * 1024, 0, args(s)); }, {}});
  }
  const uint32_t ntiles = a.nblocks / 64;
  vs.push_back({"dma persist 4w x1024WG", [&](int s) { hipLaunchKernelGGL((v_dma<4>), dim3(1024), dim3(256), 0, 0, args(s), ntiles); }, {}});
  vs.push_back({"dma persist 4w x768WG", [&](int s) { hipLaunchKernelGGL((v_dma<4>), dim3(768), dim3(256), 0, 0, args(s), ntiles); }, {}});
  vs.push_back({"dma persist 4w x512WG", [&](int s) { hipLaunchKernelGGL((v_dma<4>), dim3(512), dim3(256), 0, 0, args(s), ntiles); }, {}});
  vs.push_back({"dma persist 4w x2048WG", [&](int s) { hipLaunchKernelGGL((v_dma<4>), dim3(2048), dim3(256), 0, 0, args(s), ntiles); }, {}});
  vs.push_back({"dma oneshot 4w x4096WG", [&](int s) { hipLaunchKernelGGL((v_dma<4>), dim3(4096), dim3(256), 0, 0, args(s), ntiles); }, {}});
  vs.push_back({"compute x1 (=one plain)", [&](int s) { hipLaunchKernelGGL((v_compute<1>), dim3(a.nblocks / 256), dim3(256), 0, 0, args(s)); }, {}});
  vs.push_back({"compute x3", [&](int s) { hipLaunchKernelGGL((v_compute<3>), dim3(a.nblocks / 256), dim3(256), 0, 0, args(s)); }, {}});
  vs.push_back({"compute x5", [&](int s) { hipLaunchKernelGGL((v_compute<5>), dim3(a.nblocks / 256), dim3(256), 0, 0, args(s)); }, {}});
  for (auto &v : vs) { for (int i = 0; i < 4; i++) v.f(i % NS); }
  hipDeviceSynchronize();
  // correctness of every variant: round trip must be the identity
  for (auto &v : vs)
  {
    hipMemset(B[0], 0x55, bytes);
    v.f(0);
    std::vector<int16_t> back(W * H);
    hipMemcpy(back.data(), B[0], bytes, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (size_t i = 0; i < W * H; i++) bad += back[i] != host[i];
    if (bad) printf("!! %s: %zu mismatches\n", v.name, bad);
  }
  for (int round = 0; round < 7; round++)
    for (auto &v : vs)
    {
      hipEventRecord(e0, 0);
      for (int i = 0; i < 20; i++) v.f(i % NS);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      v.t.push_back(ms / 20);
    }
  for (auto &v : vs)
  {
    std::sort(v.t.begin(), v.t.end());
    printf("%-24s median %7.2f us  min %7.2f us   %7.1f GB/s (median)\n", v.name, v.t[v.t.size() / 2] * 1e3, v.t[0] * 1e3, 2.0 * bytes / (v.t[v.t.size() / 2] * 1e-3) / 1e9);
  }
  return 0;
}
